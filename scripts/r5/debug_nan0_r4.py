import sys, os, torch
ROOT = sys.argv[1]
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import test_gpu_parity as P
import test_gpu_parity_hard as H
from difflinker_amd import Dynamics
from oracle import egnn_oracle
from helpers import seeded_state_dict
from oracle.egnn_oracle import EGNNConfig
def case(special_is_linker, sizes, linkers):
    nf = 9
    inp, z, t = P.ragged_inputs(list(sizes), list(linkers), nf, seed=300)
    special = sizes[0] - 3 if special_is_linker else 3
    z[:, :, 3 + 7] = 0.0
    z[0, special, 3 + 7] = 1.0
    sd = seeded_state_dict(nf + 2, 128, 1, 301)
    for v in sd.values():
        v.zero_()
    sd['dynamics.embedding.weight'][0, 7] = 1e10
    sd['dynamics.e_block_0.gcl_equiv.coord_mlp.0.weight'][0, 0] = 1e10
    sd['dynamics.e_block_0.gcl_equiv.coord_mlp.2.weight'][0, 0] = 1e12
    sd['dynamics.e_block_0.gcl_equiv.coord_mlp.4.weight'][0, 0] = 1e4
    return inp, z, t, sd, EGNNConfig(in_node_nf=nf, context_node_nf=1, n_layers=1), special
print('tree:', ROOT)
for precision in ('f16x3', 'fp32'):
    inp, z, t, sd, cfg, special = case(True, (120, 12), (9, 4))
    dyn = Dynamics(n_dims=3, in_node_nf=9, context_node_nf=1, hidden_nf=128, n_layers=1, norm_constant=1e-6)
    dyn.precision = precision
    dyn.load_state_dict(sd, strict=True)
    dyn = dyn.to(P.dev())
    d = P.dev()
    prep = dyn.prepare(inp['node_mask'].to(d), inp['linker_mask'].to(d), inp['edge_mask'].to(d), inp['context'].to(d))
    out, flags = dyn._launch_forward(t.to(d), z.to(d), None, None, None, None, large=prep['large'], prep=prep)
    torch.cuda.synchronize()
    ref = egnn_oracle.dynamics_forward(sd, cfg, t, z, inp['node_mask'], inp['linker_mask'], inp['edge_mask'], inp['context'])[0, special, :3].tolist()
    print(precision, 'flags', flags.cpu().tolist(), 'vel[special]', out[0, special, :3].cpu().tolist(), 'oracle', ref)
