"""Diagnostics: randomized checks of the two small device paths beside the denoiser - the linker-size predictor (csrc/size_gnn.hip)
against oracle/size_oracle.py, and the counter-based noise (dl_philox_fill) against oracle/philox_oracle.py - on random shapes,
depths, seeds and offsets.  Expected values are made in the build container:
    python scripts/r5/fuzz_misc.py --make _fuzz/misc.pt [--cases 120]      /      python scripts/r5/fuzz_misc.py --check _fuzz/misc.pt"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import test_gpu_size_gnn as S                                                   # noqa: E402
from helpers import rel_l2, max_abs, seeded_size_state_dict                     # noqa: E402
from oracle import size_oracle, philox_oracle                                   # noqa: E402


def draw_size(seed):
    rng = np.random.default_rng(seed)
    c = dict(seed=seed, in_nf=int(rng.choice([8, 9])), out_nf=int(rng.integers(2, 40)), L=int(rng.integers(1, 5)), bn=bool(rng.random() < 0.4))
    nmol = int(rng.integers(1, 9))
    c['sizes'], c['linkers'] = [], []
    for _ in range(nmol):
        frag = int(rng.integers(0, 65)) if rng.random() < 0.8 else 64          # 0..64 fragment atoms (64: the kernel's maximum)
        link = int(rng.integers(0 if frag else 1, 12))
        c['sizes'].append(frag + link)
        c['linkers'].append(link)
    c['scale'] = float(rng.choice([0.5, 1.2, 1.6, 3.0]))                        # how many pairs fall inside the squared-distance filter
    return c


def size_case(c):
    sd = seeded_size_state_dict(c['in_nf'], 128, c['out_nf'], c['L'], seed=c['seed'], batch_norm=c['bn'], prefix='gnn.')
    data = S.random_batch(c['sizes'], c['linkers'], c['in_nf'], seed=c['seed'] + 1, scale=c['scale'])
    return sd, data


def draw_philox(seed):
    rng = np.random.default_rng(seed)
    return dict(seed=int(rng.integers(0, 2 ** 63)) * 2 + int(rng.integers(0, 2)), B=int(rng.integers(1, 7)), N=int(rng.integers(1, 60)),
                nf=int(rng.choice([8, 9])), n_draws=int(rng.integers(1, 9)), mol_offset=int(rng.choice([0, 1, 255, 256, 65535, 2 ** 31 - 3])))


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--make')
    ap.add_argument('--check')
    ap.add_argument('--cases', type=int, default=120)
    a = ap.parse_args()
    if a.make:
        store = {'size': {}, 'philox': {}}
        for k in range(a.cases):
            c = draw_size(7000 + k)
            sd, data = size_case(c)
            store['size'][7000 + k] = size_oracle.size_classifier_logits(sd, data['one_hot'], data['positions'], data['fragment_mask'],
                                                                         data['edge_mask'], c['L'], batch_norm=c['bn'])
        for k in range(a.cases // 2):
            c = draw_philox(9000 + k)
            store['philox'][9000 + k] = philox_oracle.normal_bank(c['seed'], c['B'], c['N'], c['nf'], c['n_draws'], mol_offset=c['mol_offset'])
        os.makedirs(os.path.dirname(os.path.abspath(a.make)), exist_ok=True)
        torch.save(store, a.make)
        print(f'{len(store["size"])} size-predictor cases, {len(store["philox"])} noise banks, {os.path.getsize(a.make) / 1e6:.1f} MB')
    else:
        store = torch.load(a.check, weights_only=False)
        bad, worst = [], 0.0
        for seed, ref in store['size'].items():
            c = draw_size(seed)
            sd, data = size_case(c)
            from difflinker_amd.linker_size import SizeClassifier
            clf = SizeClassifier(in_node_nf=c['in_nf'], hidden_nf=128, out_node_nf=c['out_nf'], n_layers=c['L'],
                                 normalization='batch_norm' if c['bn'] else None).eval()
            clf.load_state_dict(sd, strict=True)
            try:
                out, _ = clf.to(S.dev()).forward(S.to_dev(data), return_loss=False)
                err = rel_l2(out.cpu(), ref)
            except Exception as e:                                              # noqa: BLE001
                err = f'{type(e).__name__}: {str(e)[:160]}'
            ok = isinstance(err, float) and err <= S.TOL
            worst = max(worst, err) if isinstance(err, float) else worst
            print(('ok  ' if ok else 'FAIL'), 'size predictor', err if not isinstance(err, float) else f'{err:.2e}', c, flush=True)
            if not ok:
                bad.append(('size', c, err))
        print(f'{len(store["size"])} size-predictor cases, worst rel-L2 {worst:.2e}')
        import test_gpu_philox as X
        worst = 0.0
        for seed, (rx, rh) in store['philox'].items():
            c = draw_philox(seed)
            edm, _, _ = X.make_edm(c['nf'], 1, T=4, seed=1)
            nx, nh = edm.philox_noise_bank(c['B'], c['N'], S.dev(), mol_offset=c['mol_offset'], seed=c['seed'], n_draws=c['n_draws'])
            e = max(max_abs(nx.cpu(), torch.as_tensor(rx)), max_abs(nh.cpu(), torch.as_tensor(rh)))
            worst = max(worst, e)
            ok = e <= 4e-6 and tuple(nx.shape) == tuple(rx.shape)
            print(('ok  ' if ok else 'FAIL'), f'noise bank max-abs {e:.2e}', c, flush=True)
            if not ok:
                bad.append(('philox', c, e))
        print(f'{len(store["philox"])} noise banks, worst max-abs {worst:.2e}; {len(bad)} failures in all')
        for b in bad:
            print('FAILED:', b)
