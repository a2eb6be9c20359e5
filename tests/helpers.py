"""Shared test helpers: numpy-seeded weights (reproducible without storing them), metrics."""
import math

import numpy as np
import torch


def dynamics_param_shapes(fin, hidden_nf, n_layers, inv_sublayers=2, prefix='dynamics'):
    """(key, shape, fan_in, kind) for every tensor of the reference ``Dynamics`` state_dict
    (egnn_dynamics), in the reference's registration order (egnn.py:19-30,90-97,203-212)."""
    hn = hidden_nf
    out = []

    def lin(name, fo, fi, bias=True):
        out.append((f'{name}.weight', (fo, fi), fi, 'w'))
        if bias:
            out.append((f'{name}.bias', (fo,), fi, 'b'))

    lin(f'{prefix}.embedding', hn, fin)
    lin(f'{prefix}.embedding_out', fin, hn)
    for i in range(n_layers):
        blk = f'{prefix}.e_block_{i}'
        for j in range(inv_sublayers):
            lin(f'{blk}.gcl_{j}.edge_mlp.0', hn, 2 * hn + 2)
            lin(f'{blk}.gcl_{j}.edge_mlp.2', hn, hn)
            lin(f'{blk}.gcl_{j}.node_mlp.0', hn, 2 * hn)
            lin(f'{blk}.gcl_{j}.node_mlp.2', hn, hn)
        lin(f'{blk}.gcl_equiv.coord_mlp.0', hn, 2 * hn + 2)
        lin(f'{blk}.gcl_equiv.coord_mlp.2', hn, hn)
        out.append((f'{blk}.gcl_equiv.coord_mlp.4.weight', (1, hn), hn, 'coord'))
    return out


def seeded_state_dict(fin, hidden_nf, n_layers, seed, coord_gain=0.02, inv_sublayers=2, prefix='dynamics',
                      dtype=torch.float32):
    """Weights from ``numpy.random.default_rng(seed)`` (PCG64: stable across versions), shaped
    like ``nn.Linear``'s default init (U(-1/sqrt(fan_in), 1/sqrt(fan_in))); the coordinate head
    uses xavier-uniform with ``coord_gain`` (0.001 = reference default egnn.py:90-91; 0.02 keeps
    T=500 chains finite but lively)."""
    rng = np.random.default_rng(seed)
    sd = {}
    for key, shape, fan_in, kind in dynamics_param_shapes(fin, hidden_nf, n_layers, inv_sublayers, prefix):
        if kind == 'coord':
            bound = coord_gain * math.sqrt(6.0 / (shape[0] + shape[1]))
        else:
            bound = 1.0 / math.sqrt(fan_in)
        sd[key] = torch.from_numpy(rng.uniform(-bound, bound, size=shape).astype(np.float32)).to(dtype)
    return sd


def rel_l2(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def max_abs(a, b):
    return float((a.double() - b.double()).abs().max())
