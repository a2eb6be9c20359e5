"""CPU emulation of the f16x3 arithmetic AS THE KERNELS RUN IT - power-of-two scales taken from a-priori bounds (row-L1 norms
of the packed matrices x measured maxima), one scale per matrix / per activation tensor and molecule - inside the oracle's
forward, to find out what those scales cost on weights with the statistics of a TRAINED checkpoint
(tests/helpers.trained_like_state_dict: log-normal row factors, one row x 2^10, biases x 30), and what each candidate fix buys,
BEFORE any of it is built (round 5; the GPU test that exposed it: tests/test_gpu_round5.py).

Every 128-wide contraction of a block is emulated (operands scaled, split into fp16 hi + lo exactly as the kernels do -
truncated hi, fp16 subnormals and all - three products, fp32-class accumulation, exact rescale); everything else is the
oracle's fp32.  Error is taken against the fp64 oracle.

Switches (``--fix a,b,...``):
  rows     one power-of-two scale per OUTPUT ROW of every packed matrix instead of one per matrix
  hidden   per-hidden-feature exponents n_k on the post-activation operands of the second layers (edge / coordinate / node
           MLP): a'_k = a_k 2^n_k, W'[:, k] = W[:, k] 2^-n_k, n_k from the static magnitude proxy of feature k
  slab     the same at k-slab granularity (16 features share n after sorting the hidden features by proxy): what the pair loop
           can do with one scalar per slab
  tight_h  h fragment rows scaled by the measured max |h| instead of the bound
Run:  python scripts/numerics/emulate_bounds.py [--batch 8] [--weights trained|plain] [--fix rows,hidden]
"""
import argparse
import math
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from oracle import egnn_oracle  # noqa: E402
from difflinker_amd import synthetic  # noqa: E402
from helpers import seeded_state_dict, trained_like_state_dict, rel_l2  # noqa: E402

C = -1.4426950408889634
FIX = set()
STATE = {}


def scale_for(bound):
    """largest power of two s with bound * s < 2^15 (pack_layout.h: scale_for), clamped to 2^+-60"""
    b = torch.as_tensor(bound, dtype=torch.float64).clamp_min(1e-300)
    e = torch.floor(torch.log2(b))
    return torch.pow(2.0, (14.0 - e).clamp(-60, 60))


def f16_rtz(x):
    """fp64 -> nearest-toward-zero value on the fp16 grid (normals and subnormals; no overflow handling needed here)"""
    ax = x.abs().clamp_min(1e-300)
    ulp = torch.pow(2.0, torch.clamp(torch.floor(torch.log2(ax)) - 10.0, min=-24.0))
    return torch.trunc(x / ulp) * ulp


def split_act(xs):
    """the kernels' activation split of an already scaled fp32 value: hi = fp16(x & 0xffffe000) (truncated), lo = fp16(x - that)"""
    xs = xs.float()
    h32 = (xs.view(torch.int32) & ~0x1fff).view(torch.float32)
    return f16_rtz(h32.double()), f16_rtz((xs - h32).double())


def split_w(ws):
    """host split (split_f16): RNE hi, RNE lo"""
    ws = ws.float()
    hi = ws.half().float()
    lo = (ws - hi).half().float()
    return hi.double(), lo.double()


def w_scale(w, per_row, order=None):
    if order is not None and 'tiles' in FIX:        # 32 consecutive rows of the SORTED matrix share one scale
        m = torch.empty(w.shape[0], 1, dtype=torch.float64)
        for s0 in range(0, w.shape[0], 32):
            idx = order[s0:s0 + 32]
            m[idx] = w[idx].abs().max().double()
    else:
        m = w.abs().amax(dim=1, keepdim=True) if per_row else w.abs().max().reshape(1, 1)
    return torch.pow(2.0, (14.0 - torch.floor(torch.log2(m.double().clamp_min(1e-300)))).clamp(-60, 60))


def gemm(x, sx, w, n_k=None, order=None):
    """x [R,K] fp32 with per-row power-of-two scales sx [R,1]; w [F,K] (already folded: c, 1/norm); optional per-k exponents.
    returns x @ w.T in fp32 as the f16x3 scheme computes it"""
    w = w.double()
    x = x.double()
    if n_k is not None:
        x = x * torch.pow(2.0, n_k)[None, :]
        w = w * torch.pow(2.0, -n_k)[None, :]
    sw = w_scale(w, 'rows' in FIX, order)
    whi, wlo = split_w(w * sw)
    ahi, alo = split_act(x * sx)
    acc = ahi @ whi.t() + ahi @ wlo.t() + alo @ whi.t()
    return (acc / sx / sw.t()).float()


def silu_u(y):
    return y * torch.sigmoid(-y / C) if False else (y / (1.0 + torch.exp2(y)))


def per_mol(v, rows_per_mol):
    """max |v| per molecule -> [B]"""
    return v.abs().reshape(-1, rows_per_mol * v.shape[-1]).amax(dim=1).double()


def expand(vb, rows_per_mol):
    return vb.repeat_interleave(rows_per_mol).unsqueeze(1)


def row_l1(w):
    return float(w.double().abs().sum(1).max() * 1.0001)


def hidden_exponents(proxy, group=16):
    """n_k >= 0: how far feature k's magnitude proxy sits below the largest one (whole binades); returns (n, sort order)"""
    p = proxy.double().clamp_min(1e-300)
    n = torch.floor(torch.log2(p.max() / p))
    order = torch.argsort(p, descending=True)
    if 'slab' in FIX or 'tiles' in FIX:     # sorted by magnitude, groups of `group` share the smallest n of the group
        out = torch.empty_like(n)
        for s in range(0, n.numel(), group):
            idx = order[s:s + group]
            out[idx] = n[idx].min()
        n = out
    return n, order


def geo_term(edge_attr, wr, wd, x2, x02, N, order):
    """wr'[f] r + wd'[f] d0 as the pair loop's split-fp16 MFMA computes it: the vectors times a power of two (one per vector
    today; 'tiles': one per 32 sorted features), hi | lo; r, d0 times S1 / that, hi | lo; three products; times 1 / S1"""
    if 'exactgeo' in FIX:
        return edge_attr[:, 0:1] * wr[None, :] + edge_attr[:, 1:2] * wd[None, :]
    out = 0.0
    for col_, w_, xx in ((0, wr, x2), (1, wd, x02)):
        wcol = w_.double().reshape(-1, 1)
        sw = w_scale(wcol, False, order)                       # [1,1] or per feature [F,1]
        wv = (wcol * sw).float()
        wh = (wv.view(torch.int32) & ~0x1fff).view(torch.float32)
        whi, wlo = f16_rtz(wh.double()), f16_rtz((wv - wh).double())
        # S1: the accumulator scale; per molecule (and per feature group): scale_for(4 x2) * scale_for(max |w|) restated as scale_for(4 x2) * sw / 2^15-ish
        sx = expand(scale_for(4.0 * xx), N * N)               # [E,1] scale of r (or d0) alone
        xv = (edge_attr[:, col_:col_ + 1].double() * sx).float()
        xh = (xv.view(torch.int32) & ~0x1fff).view(torch.float32)
        xhi, xlo = f16_rtz(xh.double()), f16_rtz((xv - xh).double())
        acc = xhi @ whi.t() + xhi @ wlo.t() + xlo @ whi.t()     # [E,F]
        out = out + (acc / sx / sw.t()).float()
    return out


def edge_model(p, pre, kind, h, row, col, edge_attr, N, x2, x02, hmax, s_h):
    """first + second layer of an edge / coordinate model in the c-domain; returns y2 (pre-activation of the second layer)"""
    w1 = p[f'{pre}.{kind}.0.weight'] * C
    b1 = p[f'{pre}.{kind}.0.bias'] * C
    w2, b2 = p[f'{pre}.{kind}.2.weight'], p[f'{pre}.{kind}.2.bias'] * C
    wa, wb, wr, wd = w1[:, :128], w1[:, 128:256], w1[:, 256], w1[:, 257]
    sxh = expand(s_h, N)
    n_k, order = None, None
    if 'hidden' in FIX or 'slab' in FIX or 'tiles' in FIX:
        proxy = wa.abs().sum(1) + wb.abs().sum(1) + b1.abs() + 1e-30          # static: magnitude of feature k for |h| ~ 1
        n_k, order = hidden_exponents(proxy)
    P = gemm(h, sxh, wa, order=order) + b1
    Q = gemm(h, sxh, wb, order=order)
    D1 = P[row] + Q[col] + geo_term(edge_attr, wr, wd, x2, x02, N, order)
    pqb = (row_l1(wa) + row_l1(wb)) * hmax + float(b1.abs().max()) * 1.0001
    bound = pqb + 4.0 * (x2 * float(wr.abs().max()) + x02 * float(wd.abs().max()))
    if n_k is not None:
        # the bound of the rescaled activations (same form, per-feature maxima taken with the exponents applied)
        f = torch.pow(2.0, n_k)
        pqb = float(((wa.abs().sum(1) + wb.abs().sum(1)).double() * f).max()) * 1.0001 * hmax + float((b1.abs().double() * f).max()) * 1.0001
        bound = pqb + 4.0 * (x2 * float((wr.abs().double() * f).max()) + x02 * float((wd.abs().double() * f).max()))
    sa = expand(scale_for(bound), N * N)
    u1 = silu_u(D1)
    STATE.setdefault('log', []).append((pre + '.' + kind, float(torch.log2(bound.max())), float(u1.abs().median().clamp_min(1e-30).log2())))
    return gemm(u1, sa, w2, n_k) + b2


def gcl_emul(p, pre, h, row, col, edge_attr, node_mask, edge_mask, cfg):
    N = STATE['N']
    hmax = per_mol(h, N)
    s_h = STATE.get('s_h')
    if s_h is None or 'tight_h' in FIX:
        s_h = scale_for(hmax)
    y2 = edge_model(p, pre, 'edge_mlp', h, row, col, edge_attr, N, STATE['x2'], STATE['x02'], hmax, s_h)
    m = silu_u(y2)
    if edge_mask is not None:
        m = m * edge_mask
    agg = torch.zeros_like(h).index_add_(0, row, m)                      # c * true message sum (1/norm folded into W3b')
    aggmax = per_mol(agg, N)
    w3 = p[f'{pre}.node_mlp.0.weight']
    w3a, w3b = w3[:, :128] * C, w3[:, 128:] / cfg.normalization_factor
    b3 = p[f'{pre}.node_mlp.0.bias'] * C
    w4, b4 = p[f'{pre}.node_mlp.2.weight'] / C, p[f'{pre}.node_mlp.2.bias']
    sxh = expand(s_h, N)
    n_k, order = None, None
    if 'hidden' in FIX or 'slab' in FIX or 'tiles' in FIX:
        proxy = w3a.abs().sum(1) + w3b.abs().sum(1) + b3.abs() + 1e-30
        n_k, order = hidden_exponents(proxy, group=32)      # (per 32-feature tile when grouped: the t rows are written per tile)
    y3 = gemm(h, sxh, w3a, order=order) + b3 + gemm(agg, expand(scale_for(aggmax), N), w3b, order=order)
    t = silu_u(y3)
    y3b = row_l1(w3a) * hmax + row_l1(w3b) * aggmax + float(b3.abs().max()) * 1.0001
    if n_k is not None:
        f = torch.pow(2.0, n_k)
        y3b_s = float((w3a.abs().sum(1).double() * f).max()) * 1.0001 * hmax + float((w3b.abs().sum(1).double() * f).max()) * 1.0001 * aggmax \
            + float((b3.abs().double() * f).max()) * 1.0001
    else:
        y3b_s = y3b
    hn = h + gemm(t, expand(scale_for(y3b_s), N), w4, n_k) + b4
    if node_mask is not None:
        hn = hn * node_mask
    # |h_new| <= max|h| + L1(W4') |t| + max|b4|   (with per-feature exponents: L1 of the rescaled W4' x the rescaled bound)
    if n_k is not None:
        l1w4 = float((w4.double().abs() * torch.pow(2.0, -n_k)[None, :]).sum(1).max()) * 1.0001
        hb = hmax + l1w4 * y3b_s + float(b4.abs().max()) * 1.0001
    else:
        hb = hmax + row_l1(w4) * y3b + float(b4.abs().max()) * 1.0001
    STATE['s_h'] = scale_for(hb)
    STATE.setdefault('hlog', []).append((float(torch.log2(hb.max())), float(torch.log2(per_mol(hn, N).max()))))
    return hn


def equiv_emul(p, pre, h, x, row, col, coord_diff, edge_attr, linker_mask, node_mask, edge_mask, cfg):
    N = STATE['N']
    hmax = per_mol(h, N)
    s_h = STATE['s_h'] if 'tight_h' not in FIX else scale_for(hmax)
    x2 = (x.double() ** 2).sum(1).reshape(-1, N).amax(1)
    y2 = edge_model(p, pre, 'coord_mlp', h, row, col, edge_attr, N, x2, STATE['x02'], hmax, s_h)
    u2 = silu_u(y2)
    w7 = p[f'{pre}.coord_mlp.4.weight'] / cfg.normalization_factor / C
    s = u2 @ w7.t()
    trans = coord_diff * s
    if edge_mask is not None:
        trans = trans * edge_mask
    agg = torch.zeros_like(x).index_add_(0, row, trans)
    if linker_mask is not None:
        agg = agg * linker_mask
    x = x + agg
    if node_mask is not None:
        x = x * node_mask
    STATE['x2'] = (x.double() ** 2).sum(1).reshape(-1, N).amax(1)
    return x


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=6)
    ap.add_argument('--weights', default='trained', choices=['trained', 'plain'])
    ap.add_argument('--fix', default='')
    ap.add_argument('--layers', type=int, default=6)
    ap.add_argument('--seed', type=int, default=7)
    a = ap.parse_args()
    nf, L = 9, a.layers
    data, _ = synthetic.make_batch('C2', seed=1, batch=a.batch)
    inp = synthetic.sampler_inputs(data)
    B, N = inp['x'].shape[:2]
    g = torch.Generator().manual_seed(4)
    z = torch.cat([inp['x'], inp['h']], dim=2) * inp['fragment_mask'] + torch.randn((B, N, 3 + nf), generator=g) * inp['linker_mask']
    t = torch.full((B, 1), 0.37)
    sd = seeded_state_dict(nf + 2, 128, L, 80, coord_gain=0.02)
    if a.weights == 'trained':
        sd = trained_like_state_dict(sd, seed=a.seed)
    cfg = egnn_oracle.EGNNConfig(in_node_nf=nf, context_node_nf=1, n_layers=L)
    sd64 = {k: v.double() for k, v in sd.items()}
    ref64 = egnn_oracle.dynamics_forward(sd64, cfg, t.double(), z.double(), inp['node_mask'], inp['linker_mask'].double(),
                                         inp['edge_mask'], inp['context'].double())
    ref32 = egnn_oracle.dynamics_forward(sd, cfg, t, z, inp['node_mask'], inp['linker_mask'], inp['edge_mask'], inp['context'])
    print(f'C2-shaped batch B={B} N={N} L={L}, {a.weights} weights; fp32 oracle vs fp64: h {rel_l2(ref32[..., 3:], ref64[..., 3:]):.3e}, '
          f'vel abs {float((ref32[..., :3].double() - ref64[..., :3]).norm()):.3e}')
    orig = (egnn_oracle.gcl, egnn_oracle.equivariant_update)
    for fix in [''] + [f for f in a.fix.split(';') if f]:
        FIX.clear()
        FIX.update(x for x in fix.split(',') if x)
        STATE.clear()
        xm = (z[..., :3] * inp['node_mask']).double()
        STATE.update(N=N, x2=(xm ** 2).sum(2).amax(1), x02=(xm ** 2).sum(2).amax(1))
        egnn_oracle.gcl, egnn_oracle.equivariant_update = gcl_emul, equiv_emul
        try:
            out = egnn_oracle.dynamics_forward(sd, cfg, t, z, inp['node_mask'], inp['linker_mask'], inp['edge_mask'], inp['context'])
        finally:
            egnn_oracle.gcl, egnn_oracle.equivariant_update = orig
        print(f'  f16x3 emulation, fixes [{fix or "none: the kernels today"}]: h rel-L2 {rel_l2(out[..., 3:], ref64[..., 3:]):.3e}, '
              f'vel abs {float((out[..., :3].double() - ref64[..., :3]).norm()):.3e}')
        if not fix:
            lg = STATE['log']
            print('     edge/coordinate hidden layers: log2(bound) - log2(median |activation|), by pass: '
                  + ' '.join(f'{b - m:.0f}' for _, b, m in lg))
            print('     h rows: log2(bound) - log2(max |h_new|), by GCL: ' + ' '.join(f'{b - m:.0f}' for b, m in STATE['hlog']))


if __name__ == '__main__':
    main()
