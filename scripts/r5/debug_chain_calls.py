"""Diagnostics: the denoiser calls of one case of scripts/r5/fuzz_chain.py, one by one - the inputs the fp64 oracle's chain feeds
its denoiser in the last calls, given to the kernels as single forwards.
   --make FILE --seed-case SEED [--last 5] (build container)   /   --check FILE (GPU box)"""
import argparse
import importlib.util
import os

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
spec = importlib.util.spec_from_file_location('fuzz_chain', os.path.join(HERE, 'fuzz_chain.py'))
fc = importlib.util.module_from_spec(spec)
spec.loader.exec_module(fc)
from helpers import rel_l2                                                      # noqa: E402
from oracle import edm_oracle                                                   # noqa: E402
import test_gpu_parity as P                                                     # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--make')
ap.add_argument('--check')
ap.add_argument('--seed-case', type=int)
ap.add_argument('--last', type=int, default=5)
a = ap.parse_args()
if a.make:
    c = fc.draw(a.seed_case)
    calls = []
    orig = edm_oracle.make_dynamics_oracle

    def make(sd, cfg, prefix='dynamics'):
        den = orig(sd, cfg, prefix)

        def d2(t, xh, *args, **kw):
            out = den(t, xh, *args, **kw)
            calls.append((t.clone(), xh.clone(), out.clone()))
            return out
        return d2
    edm_oracle.make_dynamics_oracle = make
    fc.expected(c, torch.float64)
    edm_oracle.make_dynamics_oracle = orig
    calls = calls[-a.last:]
    # the fp32 oracle on the same (fp64 chain's) inputs, rounded to fp32: what the reference's arithmetic loses per call
    sd, cfg, _ = fc.weights(c)
    inp, _ = fc.inputs(c)
    den32 = orig(sd, cfg)
    rows = []
    for t, xh, out in calls:
        o32 = den32(t.float(), xh.float(), inp['node_mask'], inp['linker_mask'], inp['edge_mask'], inp['context'])
        rows.append((t.float(), xh.float(), out, o32))
    torch.save((a.seed_case, rows), a.make)
else:
    seed, rows = torch.load(a.check)
    c = fc.draw(seed)
    print(fc.describe(c))
    from difflinker_amd import Dynamics
    sd, _, flags = fc.weights(c)
    dyn = Dynamics(n_dims=3, in_node_nf=c['nf'], context_node_nf=1, hidden_nf=c['hidden'], n_layers=c['L'], inv_sublayers=c['sub'],
                   norm_constant=1e-6, **flags)
    dyn.load_state_dict(sd, strict=True)
    dyn = dyn.to(P.dev())
    dyn.team = 1
    inp, _ = fc.inputs(c)
    lm = inp['linker_mask']
    B = lm.shape[0]
    for k, (t, xh, o64, o32) in enumerate(rows):
        print(f'call {k - len(rows)} (t = {float(t.flatten()[0]):.4f}): max |x| per molecule', ['%.1e' % float(xh[b, :, :3].abs().max()) for b in range(B)],
              ' max |velocity|', ['%.1e' % float(o64[b, :, :3].abs().max()) for b in range(B)])
        print('     fp32 oracle vs fp64:    vel', ['%.1e' % rel_l2(o32[b, :, :3] * lm[b], (o64[b, :, :3] * lm[b]).float()) for b in range(B)],
              ' h', ['%.1e' % rel_l2(o32[b, :, 3:], o64[b, :, 3:].float()) for b in range(B)])
        for precision in ('fp32', 'f16x3'):
            dyn.precision = precision
            out = P.run_hip_forward(dyn, inp, xh, t)
            print(f'     kernels {precision:5s} vs fp64:   vel', ['%.1e' % rel_l2(out[b, :, :3] * lm[b], (o64[b, :, :3] * lm[b]).float()) for b in range(B)],
                  ' h', ['%.1e' % rel_l2(out[b, :, 3:], o64[b, :, 3:].float()) for b in range(B)])
