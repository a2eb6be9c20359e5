"""Diagnostics: randomized differential test of ``EDM.sample_chain`` against the oracle's chain - random batch shapes (1..8
molecules of 2..130 atoms: one compute unit, teams, the HBM-resident host loop), chain lengths, kept frames, hyper-parameters,
weights, arithmetic modes, team policy, one launch / two launches (``split_chain``).  The oracle side is CPU work and is done
where there is no GPU to pay for:
    python scripts/r5/fuzz_chain.py --make  _fuzz/chain_s1.pt --seed 1 --cases 120      # build container: expected chains
    python scripts/r5/fuzz_chain.py --check _fuzz/chain_s1.pt                           # GPU box: the kernels against them
Every case is a pure function of its seed; the file holds the expected frames only.  Not part of the test suite."""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import test_gpu_parity as P                                                     # noqa: E402
from helpers import rel_l2, max_abs, seeded_state_dict, trained_like_state_dict   # noqa: E402
from oracle import edm_oracle, egnn_oracle                                      # noqa: E402
from oracle.egnn_oracle import EGNNConfig                                       # noqa: E402


def draw(seed):
    rng = np.random.default_rng(seed)
    c = dict(seed=seed)
    c['nf'] = int(rng.choice([8, 9]))
    c['L'], c['sub'] = int(rng.integers(1, 3)), int(rng.choice([1, 2, 2]))
    c['hidden'] = int(rng.choice([64, 128, 128]))
    c['attention'], c['tanh'] = bool(rng.random() < 0.15), bool(rng.random() < 0.15)
    c['aggregation_method'] = 'mean' if rng.random() < 0.15 else 'sum'
    c['precision'] = 'f16x3' if rng.random() < 0.8 else 'fp32'
    c['trained'] = bool(rng.random() < 0.3)
    c['team'] = str(rng.choice(['1', 'auto']))
    c['split'] = bool(rng.random() < 0.6)
    c['T'] = int(rng.integers(2, 15))
    c['keep'] = int(rng.choice([1, 2, max(1, c['T'] // 2), c['T']]))
    nmol, kind = int(rng.integers(1, 9)), rng.random()
    hi = 130 if kind < 0.1 else (110 if kind < 0.35 else 55)
    if seed >= 50 * 100000:
        # --seed 50 and up: chains long enough for the static hand-over (EDM._sample_chain_fused: T >= 20, one compute unit per
        # molecule) on ragged batches of small molecules - two launches, the second on teams of two, state handed over in HBM
        c['T'], c['team'], c['split'], hi = int(rng.integers(20, 37)), '1', True, 55
        c['keep'] = int(rng.choice([1, 2, c['T'] // 3]))
        nmol = int(rng.integers(3, 15))
    c['sizes'] = [int(rng.integers(2, hi + 1)) for _ in range(nmol)]
    if rng.random() < 0.15:                                     # all the same size: no molecule finishes before another
        c['sizes'] = [c['sizes'][0]] * nmol
    c['linkers'] = [int(rng.integers(1, min(s - 1, 12) + 1)) for s in c['sizes']]
    # one case in eight: a NaN in the noise of one linker atom, in a draw the denoiser gets to see (the initial z or a step's noise)
    # - the chain must end in FoundNaNException with the oracle's index sets (generate.py:154-161 re-samples on exactly that)
    c['nan'] = None
    if rng.random() < 0.125:
        b = int(rng.integers(nmol))
        c['nan'] = (int(rng.integers(0, c['T'] + 1)), b, c['sizes'][b] - 1 - int(rng.integers(c['linkers'][b])), bool(rng.random() < 0.5))
    return c


def describe(c):
    return (f'seed {c["seed"]}: nf={c["nf"]} L={c["L"]} sub={c["sub"]} hidden={c["hidden"]} att={c["attention"]} tanh={c["tanh"]} '
            f'{c["aggregation_method"]} {c["precision"]} team={c["team"]} split={c["split"]} trained={c["trained"]} T={c["T"]} '
            f'keep={c["keep"]} sizes={c["sizes"]} linkers={c["linkers"]}'
            + (f' NaN in draw {c["nan"][0]} ({"x" if c["nan"][3] else "h"}) of atom {c["nan"][2]} of molecule {c["nan"][1]}' if c['nan'] else ''))


def weights(c):
    flags = dict(attention=c['attention'], tanh=c['tanh'], aggregation_method=c['aggregation_method'])
    sd = seeded_state_dict(c['nf'] + 2, c['hidden'], c['L'], c['seed'], inv_sublayers=c['sub'], attention=c['attention'], coord_gain=0.02)
    if c['trained']:
        sd = trained_like_state_dict(sd, c['seed'] + 1)
    cfg = EGNNConfig(in_node_nf=c['nf'], context_node_nf=1, hidden_nf=c['hidden'], n_layers=c['L'], inv_sublayers=c['sub'], **flags)
    return sd, cfg, flags


def inputs(c):
    inp, _, _ = P.ragged_inputs(c['sizes'], c['linkers'], c['nf'], seed=c['seed'] + 2)
    B, N = inp['x'].shape[:2]
    bank = edm_oracle.NoiseBank.generate(c['T'], B, N, 3, c['nf'], seed=c['seed'] + 3)
    if c['nan']:
        k, b, atom, xpart = c['nan']
        bank.draws[2 * k + (0 if xpart else 1)][b, atom, 0] = float('nan')
    return inp, bank


def expected(c, dtype=torch.float32):
    sd, cfg, _ = weights(c)
    inp, bank = inputs(c)
    sd = {k: v.to(dtype) for k, v in sd.items()}
    bank = edm_oracle.NoiseBank([d.to(dtype) for d in bank.draws])
    orc = edm_oracle.EDMOracle(edm_oracle.make_dynamics_oracle(sd, cfg), in_node_nf=c['nf'], timesteps=500)
    orc.T = c['T']
    f = {k: (v.to(dtype) if torch.is_floating_point(v) else v) for k, v in inp.items()}
    return orc.sample_chain(f['x'], f['h'], f['node_mask'], f['fragment_mask'], f['linker_mask'], f['edge_mask'],
                            f['context'], bank, keep_frames=c['keep'])


def conditioning(c, want):
    """A chain that throws its atoms far apart is ill-conditioned: how far the reference's own fp32 arithmetic is from fp64 on the
    final linker coordinates (0 for ordinary chains: not computed)."""
    if float(want[..., :3].abs().max()) < 1e4:
        return 0.0
    inp, _ = inputs(c)
    lm = inp['linker_mask']
    w64 = expected(c, torch.float64)
    return rel_l2(want[0, :, :, :3] * lm, (w64[0, :, :, :3] * lm).float())


def measured(c):
    from difflinker_amd import EDM, Dynamics
    sd, _, flags = weights(c)
    dyn = Dynamics(n_dims=3, in_node_nf=c['nf'], context_node_nf=1, hidden_nf=c['hidden'], n_layers=c['L'], inv_sublayers=c['sub'],
                   norm_constant=1e-6, **flags)
    dyn.load_state_dict(sd, strict=True)
    dyn.precision, dyn.team = c['precision'], (c['team'] if c['team'] == 'auto' else int(c['team']))
    edm = EDM(dyn.to(P.dev()), in_node_nf=c['nf'], n_dims=3, timesteps=500, noise_schedule='polynomial_2', noise_precision=1e-5,
              loss_type='l2', norm_values=[1, 4, 10]).to(P.dev())
    edm.T = c['T']
    inp, bank = inputs(c)
    g = {k: v.to(P.dev()) for k, v in inp.items()}

    def run(split):
        edm.split_chain = split
        out = edm.sample_chain(g['x'], g['h'], g['node_mask'], g['fragment_mask'], g['linker_mask'], g['edge_mask'], g['context'],
                               keep_frames=c['keep'], noise_bank=bank.stacked())
        torch.cuda.synchronize()
        return out.cpu()
    got = run(c['split'])
    return got, run(c['split']), run(not c['split']), inp


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--make')
    ap.add_argument('--check')
    ap.add_argument('--seed', type=int, default=0)
    ap.add_argument('--cases', type=int, default=100)
    a = ap.parse_args()
    t0 = time.time()
    if a.make:
        os.makedirs(os.path.dirname(os.path.abspath(a.make)), exist_ok=True)
        store = {}
        for case in range(a.cases):
            c = draw(a.seed * 100000 + case)
            try:
                want = expected(c)
                store[c['seed']] = (want, conditioning(c, want)) if bool(torch.isfinite(want).all()) else None
            except egnn_oracle.OracleNaN as e:
                store[c['seed']] = ('nan', e.x_h_nan_idx, e.only_x_nan_idx, e.only_h_nan_idx) if c['nan'] else None
            print(('made' if store[c['seed']] is not None else 'skip (oracle not finite)'), describe(c), flush=True)
        torch.save(store, a.make)
        print(f'{len(store)} cases, {sum(v is not None for v in store.values())} with a finite chain or a planted NaN, {os.path.getsize(a.make) / 1e6:.1f} MB, {time.time() - t0:.0f} s')
    else:
        store = torch.load(a.check)
        from difflinker_amd.utils import FoundNaNException
        bad, n, out_of_range, planned = [], 0, 0, 0
        for seed, want in store.items():
            c = draw(seed)
            if want is None:
                print('skip (oracle not finite)', describe(c))
                continue
            n += 1
            cond = 0.0
            if isinstance(want, tuple) and not isinstance(want[0], str):
                want, cond = want
            if isinstance(want, tuple):                                         # a planted NaN: the oracle's exception, set for set
                sets = []
                for split in (c['split'], not c['split']):
                    c2 = dict(c, split=split)
                    try:
                        measured(c2)
                        sets.append('no exception')
                    except FoundNaNException as e:
                        sets.append((e.x_h_nan_idx, e.only_x_nan_idx, e.only_h_nan_idx))
                    except Exception as e:                                      # noqa: BLE001
                        sets.append(f'{type(e).__name__}: {str(e)[:200]}')
                if sets[0] == tuple(want[1:]) and sets[1] == sets[0]:
                    print('ok   FoundNaNException', sets[0], describe(c), flush=True)
                else:
                    bad.append(f'NaN sets {sets} != oracle {want[1:]} | {describe(c)}')
                    print('FAIL', bad[-1], flush=True)
                continue
            try:
                got, again, other, inp = measured(c)
            except FoundNaNException as e:
                if e.f16_range_idx and c['precision'] != 'fp32':                # loud by design: the chain left the f16 modes' range
                    out_of_range += 1
                    print(f'range (max |x| of the oracle\'s chain {float(want[..., :3].abs().max()):.1e}): {e} | {describe(c)}', flush=True)
                    continue
                bad.append(f'FoundNaNException: {e} | {describe(c)}')
                print('FAIL', bad[-1], flush=True)
                continue
            except Exception as e:                                              # noqa: BLE001
                bad.append(f'{type(e).__name__}: {str(e)[:300]} | {describe(c)}')
                print('FAIL', bad[-1], flush=True)
                continue
            lm, fm = inp['linker_mask'], inp['fragment_mask']
            why = []
            if got.shape != want.shape:
                why.append(f'shape {tuple(got.shape)} != {tuple(want.shape)}')
            else:
                ex = rel_l2(got[0, :, :, :3] * lm, want[0, :, :, :3] * lm)
                mism = int((got[0, :, :, 3:] != want[0, :, :, 3:]).any(-1).sum())
                efr = rel_l2(got[1:], want[1:]) if got.shape[0] > 1 else 0.0
                frag = max_abs(got[0, :, :, :3] * fm, want[0, :, :, :3] * fm)
                eo = rel_l2(other[0, :, :, :3] * lm, got[0, :, :, :3] * lm)
                # (an exploding chain - some molecule's atoms fly 1e4 and more apart, the velocity grows like |x|^3.6 and every call
                # multiplies a relative difference by that power: profiles/r05/fuzz_chain.log, debug_chain_calls.log, where each
                # single forward is within 1e-7 .. 1e-6 of fp64 - is judged against ten times what the reference's own fp32
                # arithmetic loses against fp64 on it, and the two launch modes against each other by the same bar)
                bar = max(P.CHAIN_TOL, 10 * cond)
                if ex > bar or efr > bar:
                    why.append('chain error')
                if mism:
                    why.append(f'{mism} atom types differ')
                if frag > 1e-6:
                    why.append(f'fragment atoms moved by {frag:.1e}')
                if not torch.equal(got, again):
                    why.append('not repeatable bit for bit')
                if eo > (bar if cond else 1e-5) or not torch.equal(other[0, :, :, 3:], got[0, :, :, 3:]):
                    why.append(f'one launch vs two launches: {eo:.1e}')
                if c['seed'] >= 50 * 100000:
                    from difflinker_amd import edm as edm_mod
                    plan = edm_mod.split_plan(c['sizes'], c['linkers'], c['T'] + 1, torch.cuda.get_device_properties(P.dev()).multi_processor_count, c['L'], c['sub'])
                    planned += plan is not None
                line = f'x {ex:.2e} frames {efr:.2e} split-vs-not {eo:.1e}' + (f' [exploding chain: fp32 oracle {cond:.1e} from fp64]' if cond else '')
            if why:
                bad.append(f'{"; ".join(why)} | {line if got.shape == want.shape else ""} | {describe(c)}')
                print('FAIL', bad[-1], flush=True)
            else:
                print('ok  ', line, describe(c), flush=True)
        print(f'{n} chains checked in {time.time() - t0:.0f} s: {n - len(bad) - out_of_range} ok, {out_of_range} reported beyond the f16 range, {len(bad)} failures' + (f' ({planned} of the finite chains ran in two launches)' if planned else ''))
        for b in bad:
            print('FAILED:', b)
