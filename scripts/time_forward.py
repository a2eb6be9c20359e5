"""Diagnostics (GPU box): wall time of Dynamics.forward on a uniform batch (kernel A/B experiments via DIFFLINKER_HIP_LIB)."""
import argparse, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument('--n', type=int, default=50)
ap.add_argument('--batch', type=int, default=256)
ap.add_argument('--layers', type=int, default=6)
ap.add_argument('--precision', default='f16x3')
ap.add_argument('--team', default='auto', help="compute units per molecule: 'auto', 1, 2 or 4")
ap.add_argument('--iters', type=int, default=20)
ap.add_argument('--raw', action='store_true', help='time the bare launches (no flag check / host synchronisation per call)')
a = ap.parse_args()
from difflinker_amd import Dynamics, synthetic
from difflinker_amd.datasets import collate
dev = torch.device('cuda:0')
mols = synthetic.fc_molecules(a.batch, a.n, a.n, (3, 12), 9, seed=1, uniform_size=True)
inp = {k: v.to(dev) for k, v in synthetic.sampler_inputs(collate(mols)).items()}
torch.manual_seed(0)
dyn = Dynamics(3, 9, 1, hidden_nf=128, n_layers=a.layers, norm_constant=1e-6).to(dev)
dyn.precision = a.precision
dyn.team = a.team if a.team == 'auto' else int(a.team)
B, N = inp['x'].shape[:2]
z = torch.cat([inp['x'], inp['h']], 2) * inp['fragment_mask'] + torch.randn(B, N, 12, device=dev) * inp['linker_mask']
t = torch.full((B, 1), 0.5, device=dev)
args = dict(t=t, xh=z, node_mask=inp['node_mask'], linker_mask=inp['linker_mask'], edge_mask=inp['edge_mask'], context=inp['context'])
try:
    if a.raw:
        raise RuntimeError('raw launches requested')
    for _ in range(3):
        dyn.forward(**args)
except Exception as e:      # knock-out builds may produce NaNs: time the raw launch instead
    print('note:', type(e).__name__)
    dyn.forward = lambda **kw: dyn._launch_forward(kw['t'], kw['xh'], kw['node_mask'], kw['linker_mask'], kw['edge_mask'], kw['context'])
    for _ in range(3):
        dyn.forward(**args)
torch.cuda.synchronize()
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ev0.record()
for _ in range(a.iters):
    dyn.forward(**args)
ev1.record()
torch.cuda.synchronize()
print(f'{os.environ.get("DIFFLINKER_HIP_LIB", "product")}: forward (B={B}, n={a.n}, L={a.layers}, {a.precision}, team {dyn.team_for(B)}): {ev0.elapsed_time(ev1) / a.iters:.3f} ms')
