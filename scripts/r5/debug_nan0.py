import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import test_gpu_parity as P
import test_gpu_parity_hard as H
from difflinker_amd import Dynamics
from oracle import egnn_oracle
for lk in (False, True):
    for precision in ('f16x3', 'fp32'):
        for scale in (1.0, 1e-6):
            inp, z, t, sd, cfg, special = H._overflowing_head_case(lk, sizes=(120, 12), linkers=(9, 4))
            sd['dynamics.e_block_0.gcl_equiv.coord_mlp.4.weight'][0, 0] = 1e10 * scale
            dyn = Dynamics(n_dims=3, in_node_nf=9, context_node_nf=1, hidden_nf=128, n_layers=1, norm_constant=1e-6)
            dyn.precision = precision
            dyn.load_state_dict(sd, strict=True)
            dyn = dyn.to(P.dev())
            d = P.dev()
            prep = dyn.prepare(inp['node_mask'].to(d), inp['linker_mask'].to(d), inp['edge_mask'].to(d), inp['context'].to(d))
            out, flags = dyn._launch_forward(t.to(d), z.to(d), None, None, None, None, large=prep['large'], prep=prep)
            torch.cuda.synchronize()
            ref = None
            if scale != 1.0:
                try:
                    ref = egnn_oracle.dynamics_forward(sd, cfg, t, z, inp['node_mask'], inp['linker_mask'], inp['edge_mask'], inp['context'])[0, special, :3].tolist()
                except Exception as e:
                    ref = type(e).__name__
            print('special is linker' if lk else 'special is fragment', precision, 'head scale', scale, 'flags', flags.cpu().tolist(),
                  'vel[special]', out[0, special, :3].cpu().tolist(), 'oracle', ref)
