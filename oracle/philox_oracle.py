"""TEST INFRASTRUCTURE ONLY — CPU restatement of the in-kernel noise stream.

The reference draws its noise with ``torch.randn`` on the device (``utils.sample_gaussian_with_mask``,
``src/edm.py:328-345``); the drop-in keeps that stream by default.  The optional counter-based stream restated here
(``csrc/pack_layout.h: philox4x32_10 / philox_normal``) is this repository's own design: Philox4x32-10 (Salmon et al.,
"Parallel random numbers: as easy as 1, 2, 3", SC'11; the Random123 known-answer vectors are checked in
``tests/test_oracle_golden.py``), key = seed, counter = (molecule, atom, draw, component // 4), Box-Muller on
``((r >> 8) + 0.5) * 2**-24``.  Only ``tests/`` may import this module.
"""
import numpy as np
import torch

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)


def philox4x32_10(counter, key):
    """counter [..., 4] uint32, key [..., 2] uint32 (broadcastable) -> [..., 4] uint32."""
    c = [np.asarray(counter[..., i], dtype=np.uint32) for i in range(4)]
    k0 = np.asarray(key[..., 0], dtype=np.uint32)
    k1 = np.asarray(key[..., 1], dtype=np.uint32)
    with np.errstate(over='ignore'):
        for _ in range(10):
            p0 = M0 * c[0].astype(np.uint64)
            p1 = M1 * c[2].astype(np.uint64)
            n0 = (p1 >> np.uint64(32)).astype(np.uint32) ^ c[1] ^ k0
            n1 = p1.astype(np.uint32)
            n2 = (p0 >> np.uint64(32)).astype(np.uint32) ^ c[3] ^ k1
            n3 = p0.astype(np.uint32)
            c = [n0, n1, n2, n3]
            k0 = (k0 + W0).astype(np.uint32)
            k1 = (k1 + W1).astype(np.uint32)
    return np.stack(c, axis=-1)


def normal_bank(seed, B, N, nf, n_draws, mol_offset=0, draw0=0):
    """``(noise_x [n_draws,B,N,3], noise_h [n_draws,B,N,nf])`` of ``dl_philox_fill`` (fp32 arithmetic throughout)."""
    D = 3 + nf
    k, b, n, d = np.meshgrid(np.arange(n_draws), np.arange(B), np.arange(N), np.arange(D), indexing='ij')
    counter = np.stack([b + mol_offset, n, k + draw0, d >> 2], axis=-1).astype(np.uint32)
    key = np.array([seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF], dtype=np.uint32)
    r = philox4x32_10(counter, key)
    pair = (d & 2)
    r1 = np.take_along_axis(r, pair[..., None], axis=-1)[..., 0]
    r2 = np.take_along_axis(r, (pair + 1)[..., None], axis=-1)[..., 0]
    scale = np.float32(2.0 ** -24)
    u1 = ((r1 >> np.uint32(8)).astype(np.float32) + np.float32(0.5)) * scale
    u2 = ((r2 >> np.uint32(8)).astype(np.float32) + np.float32(0.5)) * scale
    rad = np.sqrt(np.float32(-2.0) * np.log(u1)).astype(np.float32)
    ang = (np.float32(2.0) * u2).astype(np.float64) * np.pi          # cospi / sinpi of an exactly representable argument
    val = np.where((d & 1) == 1, np.sin(ang), np.cos(ang)).astype(np.float32) * rad
    val = torch.from_numpy(val.astype(np.float32))
    return val[..., :3].contiguous(), val[..., 3:].contiguous()
