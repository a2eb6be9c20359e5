"""Diagnostics (GPU box): randomized differential test of ``Dynamics.forward`` / ``DynamicsWithPockets.forward`` against the oracle -
random hyper-parameters, widths, depths, molecule sizes (LDS-resident, teams, HBM-resident), linker counts, feature magnitudes,
plain and trained-like weights, both arithmetic modes.  Not part of the test suite (it is slow on the CPU side and its cases are
random); a failing case prints the seed that reproduces it, and ``--only CASE --set key=value ...`` replays it with one drawn
parameter overridden (precision=fp32, attention=False, hidden=128, trained=False, mag=1, ...) to locate what the error follows.
    python scripts/r5/fuzz_forward.py [--cases 150] [--seed 0] [--only CASE ...] [--set key=value ...]"""
import argparse
import ast
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import test_gpu_parity as P                                                     # noqa: E402
from helpers import rel_l2, seeded_state_dict, trained_like_state_dict         # noqa: E402
from oracle import egnn_oracle                                                  # noqa: E402
from oracle.egnn_oracle import EGNNConfig                                       # noqa: E402
from difflinker_amd.utils import FoundNaNException                              # noqa: E402


def draw(seed):
    """The case of ``seed``: every random choice is drawn here, so that an override changes nothing else."""
    rng = np.random.default_rng(seed)
    c = dict(seed=seed)
    c['pockets'] = bool(rng.random() < 0.2)
    c['nf'] = int(rng.choice([8, 9]))
    c['ctx'] = 2 if c['pockets'] else int(rng.choice([1, 2]))
    c['L'], c['sub'] = int(rng.integers(1, 4)), int(rng.choice([1, 2, 2, 3]))
    c['hidden'] = int(rng.choice([32, 64, 128, 128]))
    c['attention'], c['tanh'] = bool(rng.random() < 0.25), bool(rng.random() < 0.25)
    c['aggregation_method'] = 'mean' if rng.random() < 0.2 else 'sum'
    c['ct'] = bool(rng.random() < 0.8)
    c['precision'] = 'f16x3' if rng.random() < 0.75 else 'fp32'
    c['mag'] = float(10.0 ** rng.uniform(-3, 3))
    c['trained'] = bool(rng.random() < 0.4)
    c['team'] = str(rng.choice(['1', 'auto']))
    c['graph_type'] = str(rng.choice(['4A', 'FC-4A', 'FC-10A-4A'])) if c['pockets'] else 'FC'
    c['coord_gain'] = float(rng.choice([0.02, 1.0]))
    def shape():
        return (int(rng.integers(1, 3)), int(rng.integers(5, 20)), int(rng.integers(20, 120)))

    def sizes():
        nmol, big = int(rng.integers(1, 7)), rng.random()
        hi = 130 if big < 0.15 else (110 if big < 0.4 else 55)
        return [int(rng.integers(1, hi + 1)) for _ in range(nmol)]
    # (the kind the case was drawn as first, the other after it - used only under a ``--set pockets=...`` override)
    if c['pockets']:
        c['pocket_shape'], c['sizes'] = shape(), sizes()
    else:
        c['sizes'] = sizes()
        c['linkers'] = [int(rng.integers(1, max(2, min(s, 13)))) if s > 1 else 1 for s in c['sizes']]
        c['pocket_shape'] = shape()
    if c['pockets']:
        c['linkers'] = [int(rng.integers(1, max(2, min(s, 13)))) if s > 1 else 1 for s in c['sizes']]
    return c


def build(c):
    from difflinker_amd import Dynamics, DynamicsWithPockets
    flags = dict(attention=c['attention'], tanh=c['tanh'], aggregation_method=c['aggregation_method'])
    kw = dict(graph_type=c['graph_type'] if c['graph_type'] != 'FC' else 'FC-4A') if c['pockets'] else {}
    dyn = (DynamicsWithPockets if c['pockets'] else Dynamics)(
        n_dims=3, in_node_nf=c['nf'], context_node_nf=c['ctx'], hidden_nf=c['hidden'], n_layers=c['L'], inv_sublayers=c['sub'],
        condition_time=c['ct'], norm_constant=1e-6, **flags, **kw)
    sd = seeded_state_dict(c['nf'] + c['ctx'] + int(c['ct']), c['hidden'], c['L'], c['seed'], inv_sublayers=c['sub'],
                           attention=c['attention'], coord_gain=c['coord_gain'])
    if c['trained']:
        sd = trained_like_state_dict(sd, c['seed'] + 1)
    sd['dynamics.embedding.weight'] = sd['dynamics.embedding.weight'] * c['mag']
    sd['dynamics.embedding.bias'] = sd['dynamics.embedding.bias'] * c['mag']
    dyn.load_state_dict(sd, strict=True)
    dyn.precision, dyn.team = c['precision'], (c['team'] if c['team'] == 'auto' else int(c['team']))
    cfg = EGNNConfig(in_node_nf=c['nf'], context_node_nf=c['ctx'], hidden_nf=c['hidden'], n_layers=c['L'], inv_sublayers=c['sub'],
                     condition_time=c['ct'], graph_type=kw.get('graph_type', 'FC'), **flags)
    if c['pockets']:
        b, nfrag, npocket = c['pocket_shape']
        inp, z, t = P.pocket_inputs(batch=b, n_frag=nfrag, n_pocket=npocket, linker=(3, 8), nf=c['nf'], seed=c['seed'] + 2)
        fwd = egnn_oracle.dynamics_forward_pockets
    else:
        inp, z, t = P.ragged_inputs(c['sizes'], c['linkers'], c['nf'], seed=c['seed'] + 2, ctx=c['ctx'])
        fwd = egnn_oracle.dynamics_forward
    return dyn, sd, cfg, inp, z, t, fwd


def describe(case, c, inp):
    return (f'case {case} seed {c["seed"]}: {"pockets " + c["graph_type"] if c["pockets"] else "FC"} nf={c["nf"]} ctx={c["ctx"]} L={c["L"]} '
            f'sub={c["sub"]} hidden={c["hidden"]} att={c["attention"]} tanh={c["tanh"]} {c["aggregation_method"]} time={c["ct"]} '
            f'{c["precision"]} team={c["team"]} mag={c["mag"]:.2e} trained={c["trained"]} gain={c["coord_gain"]} '
            f'sizes={tuple(int(x) for x in inp["node_mask"].squeeze(-1).sum(1).tolist())}')


def run(case, c):
    """-> (verdict, line): verdict in 'ok', 'skip', 'FAIL'."""
    dyn, sd, cfg, inp, z, t, fwd = build(c)
    tag = describe(case, c, inp)
    try:
        ref = fwd(sd, cfg, t, z, inp['node_mask'], inp['linker_mask'], inp['edge_mask'], inp['context'])
    except egnn_oracle.OracleNaN:
        return 'skip', f'(oracle NaN) {tag}'
    if not bool(torch.isfinite(ref).all()):
        return 'skip', f'(oracle inf) {tag}'
    try:
        out = P.run_hip_forward(dyn.to(P.dev()), inp, z, t)
    except FoundNaNException as e:
        if e.f16_range_idx and c['precision'] != 'fp32':                        # loud by design: beyond the f16 modes' range
            return 'skip', f'(beyond the f16 range, oracle max |out| {float(ref.abs().max()):.1e}) {tag}'
        return 'FAIL', f'FoundNaNException: {e} {tag}'
    except Exception as e:                                                      # noqa: BLE001
        return 'FAIL', f'{type(e).__name__}: {str(e)[:200]} {tag}'
    eh = rel_l2(out[..., 3:], ref[..., 3:])
    vn = float(ref[..., :3].double().norm())
    floor = 4 * 2.0 ** -24 * float(z[..., :3].double().norm())
    ev = max(0.0, float((out[..., :3].double() - ref[..., :3].double()).norm()) - floor) / max(vn, 1e-30)
    tol = 1e-5 if c['precision'] == 'fp32' else 5e-6
    ok = eh <= tol and ev <= max(tol, 1e-5)
    note = ''
    if not ok:
        # an ill-conditioned case (tanh / trained-like weights at large magnitude)?  judge both against the fp64 oracle, and
        # accept the kernel when it is no further from fp64 than 3x the reference's own fp32 arithmetic is
        r64 = fwd({k: v.double() for k, v in sd.items()}, cfg, t.double(), z.double(), inp['node_mask'], inp['linker_mask'].double(),
                  inp['edge_mask'], inp['context'].double())
        hn = float(r64[..., 3:].norm())
        rh, kh = float((ref[..., 3:].double() - r64[..., 3:]).norm()) / max(hn, 1e-30), float((out[..., 3:].double() - r64[..., 3:]).norm()) / max(hn, 1e-30)
        rv = float((ref[..., :3].double() - r64[..., :3]).norm()) / max(vn, 1e-30)
        kv = max(0.0, float((out[..., :3].double() - r64[..., :3]).norm()) - floor) / max(vn, 1e-30)
        ok = kh <= 3 * rh + tol and kv <= 3 * rv + 1e-5
        note = f' [vs fp64: oracle-fp32 h {rh:.2e} vel {rv:.2e}, kernel h {kh:.2e} vel {kv:.2e}]'
    return ('ok' if ok else 'FAIL'), f'h {eh:.2e} vel {ev:.2e}{note} {tag}'


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--cases', type=int, default=150)
    ap.add_argument('--seed', type=int, default=0)
    ap.add_argument('--only', type=int, nargs='*', help='replay just these case numbers of --seed')
    ap.add_argument('--set', nargs='*', default=[], metavar='KEY=VALUE', help='override drawn parameters (python literals)')
    a = ap.parse_args()
    over = {}
    for kv in a.set:
        k, v = kv.split('=', 1)
        try:
            over[k] = ast.literal_eval(v)
        except (ValueError, SyntaxError):
            over[k] = v
    bad, count, t0 = [], dict(ok=0, skip=0, FAIL=0), time.time()
    for case in (a.only if a.only else range(a.cases)):
        c = draw(a.seed * 100000 + case)
        assert set(over) <= set(c), f'unknown keys {set(over) - set(c)}'
        c.update(over)
        verdict, line = run(case, c)
        count[verdict] += 1
        print(f'{verdict:4s}', line, flush=True)
        if verdict == 'FAIL':
            bad.append(line)
    print(f'{sum(count.values())} cases in {time.time() - t0:.0f} s: {count["ok"]} ok, {count["skip"]} skipped (oracle not finite / beyond the f16 range), {count["FAIL"]} failures')
    for line in bad:
        print('FAILED:', line)
