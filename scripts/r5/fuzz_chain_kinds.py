"""Diagnostics: the chain fuzz of ``fuzz_chain.py`` for the two other samplers of the hot path - ``InpaintingEDM.sample_chain``
(centred denoiser over the WHOLE molecule, fragments re-drawn from q every step: edm.py:549-727) and ``EDM.sample_chain`` on a
``DynamicsWithPockets`` (radius graphs rebuilt every step from the moving linker: egnn.py:471-552) - random shapes, lengths, kept
frames, hyper-parameters, weights, arithmetic modes, against the oracle's chains computed where there is no GPU to pay for:
    python scripts/r5/fuzz_chain_kinds.py --make  _fuzz/kinds_s1.pt --seed 1 --cases 80     # build container
    python scripts/r5/fuzz_chain_kinds.py --check _fuzz/kinds_s1.pt                          # GPU box
Not part of the test suite."""
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
from helpers import rel_l2, seeded_state_dict, trained_like_state_dict         # noqa: E402
from oracle import edm_oracle, egnn_oracle                                      # noqa: E402
from oracle.egnn_oracle import EGNNConfig                                       # noqa: E402


def draw(seed):
    rng = np.random.default_rng(seed)
    c = dict(seed=seed)
    c['kind'] = 'pocket' if rng.random() < 0.4 else 'inpaint'
    c['nf'] = int(rng.choice([8, 9]))
    c['L'], c['sub'] = int(rng.integers(1, 3)), int(rng.choice([1, 2, 2]))
    c['hidden'] = int(rng.choice([64, 128, 128]))
    c['attention'], c['tanh'] = bool(rng.random() < 0.15), bool(rng.random() < 0.15)
    c['aggregation_method'] = 'mean' if rng.random() < 0.15 else 'sum'
    c['precision'] = 'f16x3' if rng.random() < 0.8 else 'fp32'
    c['trained'] = bool(rng.random() < 0.3)
    c['team'] = str(rng.choice(['1', 'auto']))
    c['T'] = int(rng.integers(2, 11))
    c['keep'] = int(rng.choice([1, 2, max(1, c['T'] // 2), c['T']]))
    c['graph_type'] = str(rng.choice(['4A', 'FC-4A', 'FC-10A-4A']))
    c['pocket_shape'] = (int(rng.integers(1, 4)), int(rng.integers(4, 20)), int(rng.integers(10, 90)), int(rng.integers(2, 5)))
    nmol = int(rng.integers(1, 7))
    hi = 110 if rng.random() < 0.3 else 50
    c['sizes'] = [int(rng.integers(2, hi + 1)) for _ in range(nmol)]
    c['linkers'] = [int(rng.integers(1, min(s - 1, 12) + 1)) for s in c['sizes']]
    return c


def describe(c):
    shape = (f'batch {c["pocket_shape"][0]} x ({c["pocket_shape"][1]} fragment + {c["pocket_shape"][2]} pocket atoms, linker {c["pocket_shape"][3]}..'
             f'{c["pocket_shape"][3] + 4}) {c["graph_type"]}') if c['kind'] == 'pocket' else f'sizes={c["sizes"]} linkers={c["linkers"]}'
    return (f'seed {c["seed"]}: {c["kind"]} nf={c["nf"]} L={c["L"]} sub={c["sub"]} hidden={c["hidden"]} att={c["attention"]} tanh={c["tanh"]} '
            f'{c["aggregation_method"]} {c["precision"]} team={c["team"]} trained={c["trained"]} T={c["T"]} keep={c["keep"]} {shape}')


def parts(c):
    """-> state dict, oracle config, constructor flags, inputs, noise bank"""
    pocket = c['kind'] == 'pocket'
    ctx = 2 if pocket else 1
    flags = dict(attention=c['attention'], tanh=c['tanh'], aggregation_method=c['aggregation_method'])
    sd = seeded_state_dict(c['nf'] + ctx + 1, c['hidden'], c['L'], c['seed'], inv_sublayers=c['sub'], attention=c['attention'], coord_gain=0.02)
    if c['trained']:
        sd = trained_like_state_dict(sd, c['seed'] + 1)
    cfg = EGNNConfig(in_node_nf=c['nf'], context_node_nf=ctx, hidden_nf=c['hidden'], n_layers=c['L'], inv_sublayers=c['sub'],
                     graph_type=c['graph_type'] if pocket else 'FC', centering=not pocket, **flags)
    if pocket:
        b, nfrag, npock, lo = c['pocket_shape']
        inp, _, _ = P.pocket_inputs(batch=b, n_frag=nfrag, n_pocket=npock, linker=(lo, lo + 4), nf=c['nf'], seed=c['seed'] + 2)
    else:
        inp, _, _ = P.ragged_inputs(c['sizes'], c['linkers'], c['nf'], seed=c['seed'] + 2)
    B, N = inp['x'].shape[:2]
    pairs = c['T'] + 2 if pocket else 2 * c['T'] + 3                     # InpaintingEDM: z_T, two draws per step, two at the end
    bank = edm_oracle.NoiseBank.generate(pairs - 2, B, N, 3, c['nf'], seed=c['seed'] + 3)
    return sd, cfg, flags, inp, bank


def expected(c):
    sd, cfg, _, inp, bank = parts(c)
    den = edm_oracle.make_dynamics_oracle(sd, cfg)
    if c['kind'] == 'pocket':
        orc = edm_oracle.EDMOracle(den, in_node_nf=c['nf'], timesteps=500)
        orc.T = c['T']
        return orc.sample_chain(inp['x'], inp['h'], inp['node_mask'], inp['fragment_mask'], inp['linker_mask'], inp['edge_mask'],
                                inp['context'], bank, keep_frames=c['keep'])
    orc = edm_oracle.InpaintingEDMOracle(den, in_node_nf=c['nf'], timesteps=500)
    orc.T = c['T']
    out = orc.sample_chain(inp['x'], inp['h'], inp['node_mask'], inp['edge_mask'], inp['fragment_mask'], inp['linker_mask'],
                           inp['context'], bank, keep_frames=c['keep'])
    assert bank.pos == len(bank.draws)
    return out


def measured(c):
    from difflinker_amd import EDM, InpaintingEDM, Dynamics, DynamicsWithPockets
    sd, _, flags, inp, bank = parts(c)
    pocket = c['kind'] == 'pocket'
    kw = dict(n_dims=3, in_node_nf=c['nf'], hidden_nf=c['hidden'], n_layers=c['L'], inv_sublayers=c['sub'], norm_constant=1e-6, **flags)
    dyn = DynamicsWithPockets(context_node_nf=2, graph_type=c['graph_type'], **kw) if pocket else Dynamics(context_node_nf=1, centering=True, **kw)
    dyn.load_state_dict(sd, strict=True)
    dyn.precision, dyn.team = c['precision'], (c['team'] if c['team'] == 'auto' else int(c['team']))
    edm = (EDM if pocket else InpaintingEDM)(dyn.to(P.dev()), in_node_nf=c['nf'], n_dims=3, timesteps=500, noise_schedule='polynomial_2',
                                            noise_precision=1e-5, loss_type='l2', norm_values=[1, 4, 10]).to(P.dev())
    edm.T = c['T']
    g = {k: v.to(P.dev()) for k, v in inp.items()}

    def run():
        if pocket:
            out = edm.sample_chain(g['x'], g['h'], g['node_mask'], g['fragment_mask'], g['linker_mask'], g['edge_mask'], g['context'],
                                   keep_frames=c['keep'], noise_bank=bank.stacked())
        else:
            out = edm.sample_chain(g['x'], g['h'], g['node_mask'], g['edge_mask'], g['fragment_mask'], g['linker_mask'], g['context'],
                                   keep_frames=c['keep'], noise_bank=bank.stacked())
        torch.cuda.synchronize()
        return out.cpu()
    return run(), run(), inp


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--make')
    ap.add_argument('--check')
    ap.add_argument('--seed', type=int, default=0)
    ap.add_argument('--cases', type=int, default=80)
    a = ap.parse_args()
    t0 = time.time()
    if a.make:
        os.makedirs(os.path.dirname(os.path.abspath(a.make)), exist_ok=True)
        store = {}
        for case in range(a.cases):
            c = draw(a.seed * 100000 + case)
            try:
                want = expected(c)
                store[c['seed']] = want if bool(torch.isfinite(want).all()) and float(want[..., :3].abs().max()) < 1e4 else None
            except egnn_oracle.OracleNaN:
                store[c['seed']] = None
            print(('made' if store[c['seed']] is not None else 'skip (oracle not finite, or an exploding chain)'), describe(c), flush=True)
        torch.save(store, a.make)
        print(f'{len(store)} cases, {sum(v is not None for v in store.values())} kept, {os.path.getsize(a.make) / 1e6:.1f} MB, {time.time() - t0:.0f} s')
    else:
        store = torch.load(a.check)
        bad, n = [], 0
        for seed, want in store.items():
            c = draw(seed)
            if want is None:
                continue
            n += 1
            try:
                got, again, inp = measured(c)
            except Exception as e:                                              # noqa: BLE001
                bad.append(f'{type(e).__name__}: {str(e)[:300]} | {describe(c)}')
                print('FAIL', bad[-1], flush=True)
                continue
            nm = inp['node_mask'].float()
            moving = inp['linker_mask'] if c['kind'] == 'pocket' else nm        # inpainting re-draws the fragments too
            why = []
            if got.shape != want.shape:
                why.append(f'shape {tuple(got.shape)} != {tuple(want.shape)}')
                line = ''
            else:
                ex = rel_l2(got[0, :, :, :3] * moving, want[0, :, :, :3] * moving)
                mism = int((got[0, :, :, 3:] != want[0, :, :, 3:]).any(-1).sum())
                efr = rel_l2(got[1:], want[1:]) if got.shape[0] > 1 else 0.0
                pad = float((got * (1 - nm)).abs().max())
                if ex > P.CHAIN_TOL or efr > P.CHAIN_TOL:
                    why.append('chain error')
                if mism:
                    why.append(f'{mism} atom types differ')
                if pad != 0.0:
                    why.append(f'padding rows hold {pad:.1e}')
                if not torch.equal(got, again):
                    why.append('not repeatable bit for bit')
                line = f'x {ex:.2e} frames {efr:.2e}'
            if why:
                bad.append(f'{"; ".join(why)} | {line} | {describe(c)}')
                print('FAIL', bad[-1], flush=True)
            else:
                print('ok  ', line, describe(c), flush=True)
        print(f'{n} chains checked in {time.time() - t0:.0f} s, {len(bad)} failures')
        for b in bad:
            print('FAILED:', b)
