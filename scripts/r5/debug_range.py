"""Diagnostics (GPU box): the HBM-resident f16x3 kernels on a model with ONE dominant hidden feature, at growing magnitudes."""
import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import test_gpu_parity as P
from difflinker_amd import Dynamics
from oracle import egnn_oracle
from helpers import seeded_state_dict
from oracle.egnn_oracle import EGNNConfig
nf = 9
for sizes in ((120, 12), (40, 12)):
  for mag in (1e2, 1e4, 1e6, 1e8, 1e10):
    inp, z, t = P.ragged_inputs(list(sizes), [9, 4], nf, seed=300)
    special = sizes[0] - 3
    z[:, :, 3 + 7] = 0.0
    z[0, special, 3 + 7] = 1.0
    sd = seeded_state_dict(nf + 2, 128, 1, 301)
    for v in sd.values():
        v.zero_()
    sd['dynamics.embedding.weight'][0, 7] = mag
    sd['dynamics.e_block_0.gcl_equiv.coord_mlp.0.weight'][0, 0] = mag
    sd['dynamics.e_block_0.gcl_equiv.coord_mlp.2.weight'][0, 0] = 1e3
    sd['dynamics.e_block_0.gcl_equiv.coord_mlp.4.weight'][0, 0] = 1.0
    cfg = EGNNConfig(in_node_nf=nf, context_node_nf=1, n_layers=1)
    ref = egnn_oracle.dynamics_forward(sd, cfg, t, z, inp['node_mask'], inp['linker_mask'], inp['edge_mask'], inp['context'])[0, special, :3]
    row = []
    for precision in ('f16x3', 'fp32'):
        dyn = Dynamics(n_dims=3, in_node_nf=9, context_node_nf=1, hidden_nf=128, n_layers=1, norm_constant=1e-6)
        dyn.precision = precision
        dyn.load_state_dict(sd, strict=True)
        dyn = dyn.to(P.dev())
        d = P.dev()
        prep = dyn.prepare(inp['node_mask'].to(d), inp['linker_mask'].to(d), inp['edge_mask'].to(d), inp['context'].to(d))
        out, flags = dyn._launch_forward(t.to(d), z.to(d), None, None, None, None, large=prep['large'], prep=prep)
        torch.cuda.synchronize()
        got = out[0, special, :3].cpu()
        row.append(f'{precision} rel err {float((got - ref).norm() / ref.norm()):.2e}')
    print(sizes, f'magnitude {mag:g}: |vel| oracle {float(ref.norm()):.3e};', '; '.join(row))
