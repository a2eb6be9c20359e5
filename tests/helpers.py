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


def size_gnn_param_shapes(in_nf, hidden_nf, out_nf, n_layers, batch_norm=False):
    """(key, shape, fan_in, kind) of ``SizeGNN`` in ``state_dict`` order (linker_size.py:45-81; egnn.py:19-38)."""
    out = [('embedding_in.weight', (hidden_nf, in_nf), in_nf, 'w'), ('embedding_in.bias', (hidden_nf,), in_nf, 'b')]

    def gcl(pre):
        items = [(f'{pre}.edge_mlp.0.weight', (hidden_nf, 2 * hidden_nf + 1), 2 * hidden_nf + 1, 'w'),
                 (f'{pre}.edge_mlp.0.bias', (hidden_nf,), 2 * hidden_nf + 1, 'b'),
                 (f'{pre}.edge_mlp.2.weight', (hidden_nf, hidden_nf), hidden_nf, 'w'),
                 (f'{pre}.edge_mlp.2.bias', (hidden_nf,), hidden_nf, 'b'),
                 (f'{pre}.node_mlp.0.weight', (hidden_nf, 2 * hidden_nf), 2 * hidden_nf, 'w'),
                 (f'{pre}.node_mlp.0.bias', (hidden_nf,), 2 * hidden_nf, 'b')]
        second = 2
        if batch_norm:
            items += [(f'{pre}.node_mlp.1.{k}', (hidden_nf,), 0, k) for k in ('weight', 'bias', 'running_mean', 'running_var')]
            items += [(f'{pre}.node_mlp.1.num_batches_tracked', (), 0, 'count')]
            second = 3
        items += [(f'{pre}.node_mlp.{second}.weight', (hidden_nf, hidden_nf), hidden_nf, 'w'),
                  (f'{pre}.node_mlp.{second}.bias', (hidden_nf,), hidden_nf, 'b')]
        if batch_norm:
            items += [(f'{pre}.node_mlp.4.{k}', (hidden_nf,), 0, k) for k in ('weight', 'bias', 'running_mean', 'running_var')]
            items += [(f'{pre}.node_mlp.4.num_batches_tracked', (), 0, 'count')]
        return items

    out += gcl('gcl1')
    for i in range(n_layers - 1):
        out += gcl(f'gcl_layers.{i}')
    out += [('embedding_out.weight', (out_nf, hidden_nf), hidden_nf, 'w'), ('embedding_out.bias', (out_nf,), hidden_nf, 'b')]
    return out


def seeded_size_state_dict(in_nf, hidden_nf, out_nf, n_layers, seed, batch_norm=False, prefix=''):
    """``SizeGNN`` weights from ``numpy.random.default_rng(seed)``; BatchNorm statistics are non-trivial on purpose."""
    rng = np.random.default_rng(seed)
    sd = {}
    for key, shape, fan_in, kind in size_gnn_param_shapes(in_nf, hidden_nf, out_nf, n_layers, batch_norm):
        if kind in ('w', 'b'):
            bound = 1.0 / math.sqrt(fan_in)
            val = rng.uniform(-bound, bound, size=shape).astype(np.float32)
        elif kind == 'weight':
            val = rng.uniform(0.5, 1.5, size=shape).astype(np.float32)
        elif kind == 'bias' or kind == 'running_mean':
            val = rng.uniform(-0.3, 0.3, size=shape).astype(np.float32)
        elif kind == 'running_var':
            val = rng.uniform(0.5, 2.0, size=shape).astype(np.float32)
        else:
            val = np.asarray(7, dtype=np.int64)
        sd[prefix + key] = torch.from_numpy(np.asarray(val))
    return sd
