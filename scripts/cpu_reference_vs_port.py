"""bench.py's ``cpu_baseline`` times the ORACLE PORT (oracle/egnn_oracle.py) - the reference itself does not exist on the GPU box.
This script, run in the build container (it imports /root/reference), times the UNMODIFIED reference's ``Dynamics.forward``
(src/egnn.py:374-447) and the port side by side on the same inputs, weights and thread counts, and checks that their outputs are
the same numbers: the port is a fair stand-in for the reference as a CPU baseline ("measurement hygiene").
    PYTHONDONTWRITEBYTECODE=1 python scripts/cpu_reference_vs_port.py [--batch 32] [--forwards 2]
"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
sys.path.insert(0, '/root/reference')
sys.dont_write_bytecode = True

from src.egnn import Dynamics as RefDynamics            # noqa: E402   (the unmodified reference)
from helpers import seeded_state_dict                   # noqa: E402
from difflinker_amd import synthetic                    # noqa: E402
from oracle import egnn_oracle                          # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=32)
ap.add_argument('--forwards', type=int, default=2)
a = ap.parse_args()

data, cfg = synthetic.make_batch('C2', seed=1000, batch=a.batch)
inp = synthetic.sampler_inputs(data)
nf, ctx, L = cfg['nf'], cfg['ctx'], cfg['n_layers']
B, N = inp['x'].shape[:2]
g = torch.Generator().manual_seed(1)
z = torch.cat([inp['x'], inp['h']], dim=2) * inp['fragment_mask'] + torch.randn((B, N, 3 + nf), generator=g) * inp['linker_mask']
t = torch.full((B, 1), 0.5)
sd = seeded_state_dict(nf + ctx + 1, 128, L, 80)
ref = RefDynamics(n_dims=3, in_node_nf=nf, context_node_nf=ctx, hidden_nf=128, device='cpu', n_layers=L, attention=False,
                  tanh=False, norm_constant=1e-6, inv_sublayers=2, sin_embedding=False, normalization_factor=100,
                  aggregation_method='sum', model='egnn_dynamics', normalization='batch_norm', centering=False, graph_type='FC')
ref.load_state_dict(sd, strict=True)
ref.eval()
ocfg = egnn_oracle.EGNNConfig(in_node_nf=nf, context_node_nf=ctx, n_layers=L)


def run_ref():
    return ref.forward(t=t, xh=z, node_mask=inp['node_mask'], linker_mask=inp['linker_mask'], edge_mask=inp['edge_mask'],
                       context=inp['context'])


def run_port():
    return egnn_oracle.dynamics_forward(sd, ocfg, t, z, inp['node_mask'], inp['linker_mask'], inp['edge_mask'], inp['context'])


print(f'C2 hparams (6 blocks), the first {B} molecules of the benchmark batch (N = {N}), host: {os.cpu_count()} logical cores, '
      f'torch {torch.__version__}')
with torch.no_grad():
    o_ref, o_port = run_ref(), run_port()
    print(f'outputs: max |reference - port| = {float((o_ref - o_port).abs().max()):.3e} (rel-L2 '
          f'{float((o_ref - o_port).norm() / o_ref.norm()):.3e})')
    for threads in sorted({min(8, os.cpu_count()), min(16, os.cpu_count()), min(32, os.cpu_count())}):
        torch.set_num_threads(threads)
        row = []
        for name, fn in (('reference Dynamics.forward', run_ref), ('oracle port', run_port)):
            fn()                                        # warm-up (the reference builds and caches its edge list here)
            t0 = time.perf_counter()
            for _ in range(a.forwards):
                fn()
            row.append((name, (time.perf_counter() - t0) / a.forwards))
        print(f'{threads:3d} threads: ' + ', '.join(f'{n} {dt:.3f} s / forward' for n, dt in row) +
              f'  (port / reference = {row[1][1] / row[0][1]:.2f})')
