"""CPU emulation of candidate arithmetic schemes for the pair loop's second edge layer (the 128x128 contraction per pair
that the HIP kernels run on the matrix pipe): what each scheme costs in accuracy, BEFORE any of it is built.

The oracle (oracle/egnn_oracle.py, test infrastructure) runs a Dynamics.forward in fp32 with the product of the layers
``edge_mlp.2`` / ``coord_mlp.2`` replaced by an emulation of the scheme; the error is taken against the fp64 oracle.

  f32      plain fp32 (the oracle as it is; the floor)
  f16x3    a = hi(trunc) + lo, W = hi + lo, hi*hi' + hi*lo' + lo*hi'  on f16 products, fp32 accumulation  (the product today)
  f16x2rn  a rounded to nearest fp16, W = hi + lo: a*(hi' + lo')                                         (review item 1b)
  f8cross  hi*hi' on f16; the two cross terms with BOTH operands in fp8 e4m3 (v_mfma_f32_32x32x64_f8f6f4)  (review item 1a)
  f8cross_rn  the same with hi rounded to nearest (the lo parts half as large)
  f8half   hi*hi' and lo*hi' on f16, only hi*lo' in fp8
  bf8a     f8cross with the activation-side fp8 operands in bf8 e5m2 (loose scales are harmless there)
``--loose K`` scales the fp8 operands of the activations 2^K below the tight fit (a-priori bounds are loose).
Run:  python scripts/numerics/emulate_split.py [--batch 8] [--gain 0.001] [--loose 0]
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import egnn_oracle  # noqa: E402
from difflinker_amd import Dynamics, synthetic  # noqa: E402

F8 = torch.float8_e4m3fn
BF8 = torch.float8_e5m2


def pow2_scale(amax, top):
    """largest power of two s with amax * s <= top"""
    return 2.0 ** torch.floor(torch.log2(top / amax.clamp_min(1e-30)))


def to_f16_trunc(x):
    """fp32 -> value of the fp16 obtained by clearing the low 13 mantissa bits (round toward zero; normal range)"""
    return (x.view(torch.int32) & ~0x1fff).view(torch.float32)


def q8(x, fmt, top):
    return x.clamp(-top, top).to(fmt).to(torch.float32)


def mm(a, b):
    """products exact, accumulation fp32-class: emulate with fp64 matmul rounded once (optimistic by ~1e-7, same for all)"""
    return (a.double() @ b.double().t()).float()


class Scheme:
    def __init__(self, name, loose=0):
        self.name, self.loose = name, loose

    def linear(self, x, w, b):
        n = self.name
        if n == 'f32':
            return torch.nn.functional.linear(x, w, b)
        # power-of-two scales into the fp16 range (tight here; the kernels' bounds are looser, harmless for fp16)
        sa = pow2_scale(x.abs().max(), 32768.0)
        sw = pow2_scale(w.abs().max(), 32768.0)
        xs, ws = x * sa, w * sw
        inv = 1.0 / (sa * sw)
        w_hi_t = to_f16_trunc(ws)
        w_lo_t = (ws - w_hi_t).half().float()
        w_hi_r = ws.half().float()
        w_lo_r = (ws - w_hi_r).half().float()
        if n == 'f16x3':
            hi = to_f16_trunc(xs); lo = (xs - hi).half().float()
            acc = mm(hi, w_hi_t) + mm(hi, w_lo_t) + mm(lo, w_hi_t)
        elif n == 'f16x2rn':
            a = xs.half().float()
            acc = mm(a, w_hi_t) + mm(a, w_lo_t)
        elif n in ('f8cross', 'f8cross_rn', 'bf8a', 'f8half'):
            rn = n != 'f8cross'
            hi = xs.half().float() if rn else to_f16_trunc(xs)
            lo = xs - hi
            w_hi, w_lo = (w_hi_r, ws - w_hi_r) if rn else (w_hi_t, ws - w_hi_t)
            fa, ta = (BF8, 57344.0) if n == 'bf8a' else (F8, 448.0)
            # activation-side fp8 operands: one scale per pass (tight fit / 2^loose)
            s_a8 = pow2_scale(xs.abs().max(), ta) * 2.0 ** -self.loose
            s_l8 = pow2_scale(lo.abs().max(), ta) * 2.0 ** -self.loose
            a8 = q8(xs * s_a8, fa, ta) / s_a8
            l8 = q8(lo * s_l8, fa, ta) / s_l8
            # weight-side fp8 operands: static, one scale per output row (MX block scales would be finer still)
            s_w8 = pow2_scale(ws.abs().amax(dim=1, keepdim=True), 448.0)
            s_wl8 = pow2_scale(w_lo.abs().amax(dim=1, keepdim=True), 448.0)
            w8 = q8(ws * s_w8, F8, 448.0) / s_w8
            wl8 = q8(w_lo * s_wl8, F8, 448.0) / s_wl8
            if n == 'f8half':
                acc = mm(hi, w_hi) + mm(lo.half().float(), w_hi) + mm(a8, wl8)
            else:
                acc = mm(hi, w_hi) + mm(l8, w8) + mm(a8, wl8)
        else:
            raise ValueError(n)
        return acc * inv + b


def run(args):
    torch.manual_seed(0)
    data, cfg = synthetic.make_batch('C2', seed=3, batch=args.batch)
    inp = synthetic.sampler_inputs(data)
    nf, ctx, L = cfg['nf'], cfg['ctx'], cfg['n_layers']
    dyn = Dynamics(n_dims=3, in_node_nf=nf, context_node_nf=ctx, hidden_nf=128, n_layers=L, norm_constant=1e-6)
    if args.gain != 0.001:
        for k, p in dyn.named_parameters():
            if k.endswith('coord_mlp.4.weight'):
                torch.nn.init.xavier_uniform_(p.data, gain=args.gain)
    sd = {k: v.detach().clone() for k, v in dyn.state_dict().items()}
    ocfg = egnn_oracle.EGNNConfig(in_node_nf=nf, context_node_nf=ctx, n_layers=L)
    B, N = inp['x'].shape[:2]
    g = torch.Generator().manual_seed(11)
    z = torch.cat([inp['x'], inp['h']], 2) * inp['fragment_mask'] + torch.randn(B, N, 3 + nf, generator=g) * inp['linker_mask']
    t = torch.rand(B, 1, generator=g)

    def forward(sd_, dtype):
        c = lambda v: v.to(dtype) if v.is_floating_point() else v
        return egnn_oracle.dynamics_forward({k: c(v) for k, v in sd_.items()}, ocfg, c(t), c(z), inp['node_mask'],
                                            c(inp['linker_mask']), inp['edge_mask'], c(inp['context']))

    truth = forward(sd, torch.float64)
    ref32 = forward(sd, torch.float32).double()        # what the GPU parity tests compare with
    orig = egnn_oracle._lin
    print(f'C2-shaped batch B={B} N={N} L={L}, coordinate-head gain {args.gain}, fp8 activation scales 2^-{args.loose} of tight')
    print(f'{"scheme":12s} {"rel-L2 h":>10s} {"rel-L2 vel":>11s} | vs the fp32 oracle: {"h":>9s} {"vel":>9s}')
    for name in args.schemes.split(','):
        sch = Scheme(name, args.loose)
        coord = Scheme(args.coord or name, args.loose)

        def lin(p, key, x, sch=sch, coord=coord):
            if x.dtype == torch.float32 and key.endswith('edge_mlp.2'):
                return sch.linear(x, p[key + '.weight'], p[key + '.bias'])
            if x.dtype == torch.float32 and key.endswith('coord_mlp.2'):       # --coord: the coordinate passes' scheme
                return coord.linear(x, p[key + '.weight'], p[key + '.bias'])
            return orig(p, key, x)
        egnn_oracle._lin = lin
        try:
            out = forward(sd, torch.float32).double()
        finally:
            egnn_oracle._lin = orig
        eh = float((out[..., 3:] - truth[..., 3:]).norm() / truth[..., 3:].norm())
        ev = float((out[..., :3] - truth[..., :3]).norm() / truth[..., :3].norm())
        eh32 = float((out[..., 3:] - ref32[..., 3:]).norm() / ref32[..., 3:].norm())
        ev32 = float((out[..., :3] - ref32[..., :3]).norm() / ref32[..., :3].norm())
        print(f'{name:12s} {eh:10.2e} {ev:11.2e} | {"":19s} {eh32:9.2e} {ev32:9.2e}')


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=8)
    ap.add_argument('--gain', type=float, default=0.001)
    ap.add_argument('--loose', type=int, default=0)
    ap.add_argument('--coord', default='', help='scheme of the coordinate passes (default: the same as the GCL passes)')
    ap.add_argument('--schemes', default='f32,f16x3,f16x2rn,f8cross,f8cross_rn,bf8a,f8half')
    run(ap.parse_args())
