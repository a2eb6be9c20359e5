import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import test_gpu_parity as T
from oracle import edm_oracle
from difflinker_amd import EDM
from helpers import rel_l2
nf, Tn, keep = 8, 6, 2
sizes, linkers = [20, 70, 35, 60, 12, 56], [4, 9, 5, 8, 3, 6]
for team in ('auto', 1):
  for prec in ('f16x3', 'fp32'):
    dyn, sd, cfg = T.make_dynamics(nf, 1, 1, seed=33, precision=prec)
    dyn.team = team
    inp, _, _ = T.ragged_inputs(sizes, linkers, nf, seed=34)
    B, N = inp['x'].shape[:2]
    edm = EDM(dyn, in_node_nf=nf, n_dims=3, timesteps=500, noise_schedule='polynomial_2', noise_precision=1e-5, loss_type='l2', norm_values=[1, 4, 10]).to(T.dev())
    edm.T = Tn
    bank = edm_oracle.NoiseBank.generate(Tn, B, N, 3, nf, seed=35)
    orc = edm_oracle.EDMOracle(edm_oracle.make_dynamics_oracle(sd, cfg), in_node_nf=nf, timesteps=500)
    orc.T = Tn
    want = orc.sample_chain(inp['x'], inp['h'], inp['node_mask'], inp['fragment_mask'], inp['linker_mask'], inp['edge_mask'], inp['context'], bank, keep_frames=keep)
    g = {k: v.to(T.dev()) for k, v in inp.items()}
    got = edm.sample_chain(g['x'], g['h'], g['node_mask'], g['fragment_mask'], g['linker_mask'], g['edge_mask'], g['context'], keep_frames=keep, noise_bank=bank.stacked()).cpu()
    for b in range(B):
        print(team, prec, 'mol', b, sizes[b], 'frame1 x', rel_l2(got[1, b, :, :3], want[1, b, :, :3]), 'h', rel_l2(got[1, b, :, 3:], want[1, b, :, 3:]), 'frame0 x', rel_l2(got[0, b, :, :3], want[0, b, :, :3]))
