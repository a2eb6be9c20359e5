"""Diagnostics (GPU box): one-layer Dynamics.forward vs the CPU oracle for single molecules of various sizes; prints the
per-atom error pattern.  python scripts/debug_sizes.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import test_gpu_parity as T  # noqa: E402
from oracle import egnn_oracle  # noqa: E402

sizes = [int(a) for a in sys.argv[1:]] or [5, 8, 16, 17, 32, 33, 40, 50, 55]
for prec in ('fp32', 'f16x3'):
    for n in sizes:
        for L in (1, 2):
            dyn, sd, cfg = T.make_dynamics(9, 1, L, seed=100 + L, precision=prec)
            inp, z, t = T.ragged_inputs([n], [max(1, n // 6)], 9, seed=n)
            ref = egnn_oracle.dynamics_forward(sd, cfg, t, z, inp['node_mask'], inp['linker_mask'], inp['edge_mask'], inp['context'])
            out = T.run_hip_forward(dyn, inp, z, t)
            eh = float((out[..., 3:] - ref[..., 3:]).norm() / ref[..., 3:].norm())
            ev = float((out[..., :3] - ref[..., :3]).norm() / max(float(ref[..., :3].norm()), 1e-30))
            per_atom = (out[0, :, 3:] - ref[0, :, 3:]).norm(dim=1) / ref[0, :, 3:].norm(dim=1).clamp_min(1e-20)
            bad = [i for i in range(n) if per_atom[i] > 1e-4]
            print(f'{prec} n={n} L={L}: h {eh:.2e} vel {ev:.2e}  bad atoms {bad[:20]}{"..." if len(bad) > 20 else ""} ({len(bad)})', flush=True)
