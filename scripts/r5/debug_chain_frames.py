"""Diagnostics: one case of scripts/r5/fuzz_chain.py frame by frame - where along the chain, and in which molecule, the kernels
leave the oracle.   --make FILE --seed-case SEED (build container: the oracle's frames in fp32 and fp64)   /   --check FILE (GPU box)"""
import argparse
import importlib.util
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
spec = importlib.util.spec_from_file_location('fuzz_chain', os.path.join(HERE, 'fuzz_chain.py'))
fc = importlib.util.module_from_spec(spec)
spec.loader.exec_module(fc)
from helpers import rel_l2                                                      # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--make')
ap.add_argument('--check')
ap.add_argument('--seed-case', type=int, nargs='*', default=[])
a = ap.parse_args()
if a.make:
    store = {}
    for seed in a.seed_case:
        c = fc.draw(seed)
        c['keep'] = c['T']
        store[seed] = (fc.expected(c), fc.expected(c, torch.float64))
    torch.save(store, a.make)
else:
    for seed, (w32, w64) in torch.load(a.check).items():
        c = fc.draw(seed)
        c['keep'] = c['T']
        print(fc.describe(c))
        inp, _ = fc.inputs(c)
        lm = inp['linker_mask']
        for precision in ('fp32', 'f16x3'):
            for split in (False, True):
                c2 = dict(c, precision=precision, split=split)
                got, _, _, _ = fc.measured(c2)
                print(f'  {precision}, {"two launches" if split else "one launch"}: linker-x rel-L2 per molecule against the fp64 oracle (the fp32 oracle\'s in brackets)')
                for f in reversed(range(got.shape[0])):
                    row = []
                    for b in range(got.shape[1]):
                        e = rel_l2(got[f, b, :, :3] * lm[b], (w64[f, b, :, :3] * lm[b]).float())
                        o = rel_l2(w32[f, b, :, :3] * lm[b], (w64[f, b, :, :3] * lm[b]).float())
                        row.append(f'{e:.1e} ({o:.0e})')
                    print(f'    frame {f:2d}: ' + '  '.join(row))
