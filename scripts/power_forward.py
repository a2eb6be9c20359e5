"""Diagnostics (GPU box): forward time AND average board power of Dynamics.forward on a uniform batch (energy experiments;
kernel variants via DIFFLINKER_HIP_LIB).  Polls rocm-smi from a thread while the launches run for ~3 s."""
import argparse, os, re, subprocess, sys, threading, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument('--n', type=int, default=50)
ap.add_argument('--batch', type=int, default=256)
ap.add_argument('--seconds', type=float, default=3.0)
ap.add_argument('--precision', default='f16x3')
a = ap.parse_args()
from difflinker_amd import Dynamics, synthetic
from difflinker_amd.datasets import collate
dev = torch.device('cuda:0')
mols = synthetic.fc_molecules(a.batch, a.n, a.n, (3, 12), 9, seed=1, uniform_size=True)
inp = {k: v.to(dev) for k, v in synthetic.sampler_inputs(collate(mols)).items()}
torch.manual_seed(0)
dyn = Dynamics(3, 9, 1, hidden_nf=128, n_layers=6, norm_constant=1e-6).to(dev)
dyn.precision = a.precision
B, N = inp['x'].shape[:2]
z = torch.cat([inp['x'], inp['h']], 2) * inp['fragment_mask'] + torch.randn(B, N, 12, device=dev) * inp['linker_mask']
t = torch.full((B, 1), 0.5, device=dev)
launch = lambda: dyn._launch_forward(t, z, inp['node_mask'], inp['linker_mask'], inp['edge_mask'], inp['context'])
for _ in range(5):
    launch()
torch.cuda.synchronize()
samples, stop = [], [False]
def poll():
    while not stop[0]:
        out = subprocess.run(['rocm-smi', '--showpower', '--showclocks'], capture_output=True, text=True).stdout
        p = re.search(r'Power \(W\): ([0-9.]+)', out)
        c = re.search(r'sclk clock level: \S+ \((\d+)Mhz\)', out)
        if p and c:
            samples.append((float(p.group(1)), int(c.group(1))))
th = threading.Thread(target=poll); th.start()
n = 0
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ev0.record()
t0 = time.time()
while time.time() - t0 < a.seconds:
    for _ in range(20):
        launch()
    n += 20
    torch.cuda.synchronize()
ev1.record(); torch.cuda.synchronize()
stop[0] = True; th.join()
ms = ev0.elapsed_time(ev1) / n
mid = samples[1:-1] or samples
pw = sum(s[0] for s in mid) / max(1, len(mid)); ck = sum(s[1] for s in mid) / max(1, len(mid))
print(f'{os.environ.get("DIFFLINKER_HIP_LIB", "product"):>22s} {a.precision} B={B:3d}: forward {ms:.3f} ms, power {pw:6.0f} W, sclk {ck:5.0f} MHz, '
      f'energy/forward {pw * ms / 1e3:.3f} J ({len(mid)} samples)')
