"""Diagnostics: per-phase shader-clock timeline of one Dynamics.forward (workgroup 0), using the
dl_set_profile_buffer hook.  Run on the GPU box:  python scripts/phase_timeline.py [--n 50] [--batch 256]"""
import argparse
import collections
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

ap = argparse.ArgumentParser()
ap.add_argument('--n', type=int, default=50)
ap.add_argument('--batch', type=int, default=256)
ap.add_argument('--layers', type=int, default=6)
ap.add_argument('--team', default='auto', help="compute units per molecule: 'auto', 1, 2 or 4")
a = ap.parse_args()

import subprocess
import __graft_entry__ as entry
# the phase log is compiled out of the product library: build (or reuse) a -DDL_PROFILE copy under build/
if 'DIFFLINKER_HIP_LIB' not in os.environ:
    prof_lib = os.path.join(ROOT, 'build', 'libdifflinker_hip_profile.so')
    if not os.path.exists(prof_lib) or any(os.path.getmtime(src) > os.path.getmtime(prof_lib) for src in entry.HIP_SOURCES):
        os.makedirs(os.path.dirname(prof_lib), exist_ok=True)
        subprocess.run([entry.HIPCC, '--offload-arch=gfx950', '-O3', '-std=c++17', '-shared', '-fPIC', '-Wno-unused-value',
                        '-DDL_PROFILE', '-I', os.path.join(ROOT, 'include')] + entry.EXTRA_FLAGS['egnn_fc.hip'] +
                       entry.HIP_SOURCES + ['-o', prof_lib], check=True)
    os.environ['DIFFLINKER_HIP_LIB'] = prof_lib
entry.build()
from difflinker_amd import Dynamics, synthetic, _lib

dev = torch.device('cuda:0')
lib = _lib.load()
mols = synthetic.fc_molecules(a.batch, a.n, a.n, (3, 12), 9, seed=1, uniform_size=True)
from difflinker_amd.datasets import collate
inp = synthetic.sampler_inputs(collate(mols))
inp = {k: v.to(dev) for k, v in inp.items()}
torch.manual_seed(0)
dyn = Dynamics(3, 9, 1, hidden_nf=128, n_layers=a.layers, norm_constant=1e-6).to(dev)
dyn.team = a.team if a.team == 'auto' else int(a.team)
B, N = inp['x'].shape[:2]
z = torch.cat([inp['x'], inp['h']], 2) * inp['fragment_mask'] + torch.randn(B, N, 12, device=dev) * inp['linker_mask']
t = torch.full((B, 1), 0.5, device=dev)
args = dict(t=t, xh=z, node_mask=inp['node_mask'], linker_mask=inp['linker_mask'], edge_mask=inp['edge_mask'],
            context=inp['context'])
try:
    for _ in range(3):
        dyn.forward(**args)
except Exception as e:      # knock-out builds produce NaNs: the raw launch instead
    print('note:', type(e).__name__)
    dyn.forward = lambda **kw: dyn._launch_forward(kw['t'], kw['xh'], kw['node_mask'], kw['linker_mask'], kw['edge_mask'], kw['context'])
    for _ in range(3):
        dyn.forward(**args)
torch.cuda.synchronize()
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ev0.record()
for _ in range(5):
    dyn.forward(**args)
ev1.record()
torch.cuda.synchronize()
print(f'forward (B={B}, n={a.n}, L={a.layers}, team {dyn.team_for(B)}): {ev0.elapsed_time(ev1) / 5:.3f} ms')

maxev = lib.dl_profile_max_events()
assert maxev > 0, 'this library was built without -DDL_PROFILE'
buf = torch.zeros((8, maxev, 2), dtype=torch.int64, device=dev)
lib.dl_set_profile_buffer(ctypes.c_void_p(buf.data_ptr()))
dyn.forward(**args)
torch.cuda.synchronize()
lib.dl_set_profile_buffer(None)
ev = buf.cpu()
names = {(12, 120): 'gcl: PAIR loop, this wave', (120, 13): 'gcl: wait for the other waves + partials', (32, 121): 'eq: PAIR loop, this wave',
         (121, 33): 'eq: wait for the other waves + partials', (1, 2): 'embedding', (10, 11): 'gcl: stage+proj P,Q', (11, 12): 'gcl: barrier', (12, 13): 'gcl: PAIR loop + partials',
         (13, 14): 'gcl: reduce + h->lds', (13, 20): 'gcl: barrier (partials complete)', (20, 21): 'gcl: sum of slot partials',
         (21, 22): 'gcl: barrier (partials read)', (22, 23): 'gcl: h rows + W2\' DMA issue, agg store, max', (23, 25): 'gcl: wait for the h rows (LDS-DMA)', (25, 14): 'gcl: barrier (h, agg in place)', (20, 23): 'gcl: team exchange', (14, 15): 'gcl: node mlp 1', (15, 16): 'gcl: barrier+node mlp 2',
         (16, 10): 'gcl: end barrier', (16, 30): 'gcl: end barrier', (30, 31): 'eq: stage+proj', (31, 32): 'eq: barrier',
         (32, 33): 'eq: PAIR loop + partials', (33, 34): 'eq: reduce + x', (34, 10): 'eq: end barrier', (34, 3): 'eq: end barrier',
         (2, 10): 'h->regs', (3, 4): 'output head', (14, 104): 'gcl: mlp1 frags+gemm(h)', (104, 105): 'gcl: mlp1 image DMA issue + gemm(agg)',
         (105, 15): 'gcl: mlp1 epilogue', (14, 105): 'gcl: mlp1 frags + both gemms', (15, 106): 'gcl: barrier + mlp2 residual loads', (106, 107): 'gcl: mlp2 gemm',
         (107, 16): 'gcl: mlp2 epilogue (LDS + HBM rows)', (10, 110): 'open: projections P, Q, T0', (110, 111): 'open: barrier + exchange stores',
         (111, 112): 'open: team sync (drain, barrier, flags)', (112, 11): 'open: exchange loads -> LDS', (22, 23): 'gcl: W2\' DMA issue + agg fragment rows', (23, 14): 'gcl: barrier (agg in place)'}
# per-atom phases version 3 (stream_phase): 200 = stream begins, 210 + C = chunk C landed and published, 240 = last chunk computed,
# 250 = ring done
for pre_, nm_ in ((22, 'v3 post: agg rows, scales, first DMA issue'), (10, 'v3 open: first DMA issue, h -> fragments'), (2, 'v3 entry: first DMA issue, h -> fragments')):
    names[(pre_, 200)] = nm_
names[(200, 210)] = 'v3: wait for chunk 0 (+ barrier)'
for c_ in range(10):
    names[(210 + c_, 211 + c_)] = f'v3: chunk {c_} (mfma + epilogue; loaders: wait) + barrier'
    names[(210 + c_, 240)] = f'v3: last chunk ({c_})'
names[(240, 250)] = 'v3: ring-done barrier'
names[(250, 11)] = 'v3: P, Q rows -> LDS'
names[(250, 110)] = 'v3: P, Q rows -> LDS'
names[(11, 32)] = 'gcl: barrier'
inloop = {(40, 41): 'L1 (geo, SiLU, split)', (41, 42): 'M0 (48 mfma)', (42, 43): 'E0 (epilogue)', (43, 44): 'M1 (48 mfma)',
          (44, 45): 'E1 (epilogue)'}
for w in range(8):
    e = ev[w]
    n = int((e[:, 0] != 0).sum())
    tags = [int(x) for x in e[:n, 0]]
    ts = [int(x) for x in e[:n, 1]]
    if 200 in tags:
        # raw event log of the stream phases of passes 0..2 (tag:ticks since the phase's first event)
        seen = 0
        k = 0
        while k < n and seen < (4 if w in (0, 4) else 2):
            if tags[k] == 200:
                j = k
                while j + 1 < n and tags[j + 1] >= 200:
                    j += 1
                print(f'   wave {w} stream phase {seen}: ' + ' '.join(f'{tags[i]}:{ts[i] - ts[k]}' for i in range(k, j + 1)) + f' | next {tags[j + 1] if j + 1 < n else None}:{(ts[j + 1] - ts[k]) if j + 1 < n else None}')
                seen += 1
                k = j
            k += 1
    # (the fine-grained tags 220.. / 230.. are for the raw log only)
    keep = [i for i in range(n) if not (220 <= tags[i] < 240)]
    tags, ts = [tags[i] for i in keep], [ts[i] for i in keep]
    n = len(tags)
    # pass-level events only (tags < 40), in-loop events (40..45) of step 2 reported separately per pass kind
    top = [(tg, t_) for tg, t_ in zip(tags, ts) if tg < 40 or tg >= 100]
    tot = collections.OrderedDict()
    for k in range(len(top) - 1):
        nm = names.get((top[k][0], top[k + 1][0]), str((top[k][0], top[k + 1][0])))
        tot[nm] = tot.get(nm, 0) + top[k + 1][1] - top[k][1]
    total = top[-1][1] - top[0][1]
    loop = {'gcl': collections.OrderedDict(), 'eq': collections.OrderedDict()}
    cnt = {'gcl': 0, 'eq': 0}
    kind = 'gcl'
    for k in range(n - 1):
        if tags[k] == 12: kind = 'gcl'
        if tags[k] == 32: kind = 'eq'
        key = (tags[k], tags[k + 1])
        if key in inloop:
            loop[kind][inloop[key]] = loop[kind].get(inloop[key], 0) + ts[k + 1] - ts[k]
            if key == (40, 41): cnt[kind] += 1
    # -DDL_PROFILE_LOOP builds: 50 = a step begins, 51 = k-slab 0 produced, 52 = the matrix section is over, 53 = the epilogue is over
    step = {'gcl': collections.OrderedDict(), 'eq': collections.OrderedDict()}
    nstep = {'gcl': 0, 'eq': 0}
    kind = 'gcl'
    stepnames = {(50, 51): 'open + k-slab 0', (51, 52): 'matrix section (MFMAs + first layer)', (52, 53): 'epilogue', (53, 50): 'between steps'}
    for k in range(n - 1):
        if tags[k] == 12: kind = 'gcl'
        if tags[k] == 32: kind = 'eq'
        key = (tags[k], tags[k + 1])
        if key in stepnames:
            step[kind][stepnames[key]] = step[kind].get(stepnames[key], 0) + ts[k + 1] - ts[k]
            if key == (50, 51): nstep[kind] += 1
    if w in (0, 3, 4, 7):
        for kind_ in ('gcl', 'eq'):
            if nstep[kind_]:
                print(f'   wave {w}: {nstep[kind_]} {kind_} steps, ticks per step: ' +
                      ', '.join(f'{nm} {v // nstep[kind_]}' for nm, v in step[kind_].items()) + f' | sum {sum(step[kind_].values()) // nstep[kind_]}')
    segs = [(tg, t_) for tg, t_ in zip(tags, ts) if 40 <= tg < 50]
    if segs and w in (0, 4):
        # ping-pong builds: (40+k) = segment k of steps 2..3 finished, (60+k) = barrier k released; first pass only
        first = []
        for tg, t_ in segs:
            if first and tg == 40 and any(x[0] == 40 for x in first):
                break
            first.append((tg, t_))
        line = ' '.join(f'{tg}:{t_ - first[0][1]}' for tg, t_ in first)
        print(f'   wave {w} segment log (tag:ticks since first): {line}')
    # per-atom time: from the end of a pair loop (13 / 33: partials written) to the start of the next (12 / 32) or the head (3)
    per_atom = {'after gcl': [], 'after eq': []}
    start = None
    for tg, t_ in zip(tags, ts):
        if tg in (13, 33):
            start = (tg, t_)
        elif tg in (12, 32, 3) and start is not None:
            per_atom['after gcl' if start[0] == 13 else 'after eq'].append(t_ - start[1])
            start = None
    if w in (0, 4):
        for k_, v_ in per_atom.items():
            if v_:
                print(f'   wave {w} per-atom phases {k_}: {len(v_)} x mean {sum(v_) // len(v_)} ticks (min {min(v_)}, max {max(v_)}); sum {sum(v_)} = {100.0 * sum(v_) / max(total, 1):.1f} % of the forward')
    print(f'wave {w}: gcl PAIR {tot.get("gcl: PAIR loop + partials", 0):9d}  eq PAIR {tot.get("eq: PAIR loop + partials", 0):9d}  total {total}')
    if w in (0, 4, 7):
        print(f'--- wave {w}: {n} events, total {total} ticks')
        for nm, v in tot.items():
            print(f'   {nm:32s} {v:10d}  {100.0 * v / total:5.1f}%')
        for kind in ('gcl', 'eq'):
            if cnt[kind]:
                print(f'   in-loop, step 2 of {cnt[kind]} {kind} passes (ticks per step): ' +
                      ', '.join(f'{nm} {v // cnt[kind]}' for nm, v in loop[kind].items()) +
                      f' | sum {sum(loop[kind].values()) // cnt[kind]}')
