"""Shared test helpers: numpy-seeded weights (reproducible without storing them), metrics."""
import math

import numpy as np
import torch


def dynamics_param_shapes(fin, hidden_nf, n_layers, inv_sublayers=2, prefix='dynamics', attention=False, edge_feat_nf=2):
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
            lin(f'{blk}.gcl_{j}.edge_mlp.0', hn, 2 * hn + edge_feat_nf)
            lin(f'{blk}.gcl_{j}.edge_mlp.2', hn, hn)
            lin(f'{blk}.gcl_{j}.node_mlp.0', hn, 2 * hn)
            lin(f'{blk}.gcl_{j}.node_mlp.2', hn, hn)
            if attention:                                   # registered after node_mlp (egnn.py:42-43)
                lin(f'{blk}.gcl_{j}.att_mlp.0', 1, hn)
        lin(f'{blk}.gcl_equiv.coord_mlp.0', hn, 2 * hn + edge_feat_nf)
        lin(f'{blk}.gcl_equiv.coord_mlp.2', hn, hn)
        out.append((f'{blk}.gcl_equiv.coord_mlp.4.weight', (1, hn), hn, 'coord'))
    return out


def seeded_state_dict(fin, hidden_nf, n_layers, seed, coord_gain=0.02, inv_sublayers=2, prefix='dynamics',
                      dtype=torch.float32, attention=False, edge_feat_nf=2):
    """Weights from ``numpy.random.default_rng(seed)`` (PCG64: stable across versions), shaped
    like ``nn.Linear``'s default init (U(-1/sqrt(fan_in), 1/sqrt(fan_in))); the coordinate head
    uses xavier-uniform with ``coord_gain`` (0.001 = reference default egnn.py:90-91; 0.02 keeps
    T=500 chains finite but lively)."""
    rng = np.random.default_rng(seed)
    sd = {}
    for key, shape, fan_in, kind in dynamics_param_shapes(fin, hidden_nf, n_layers, inv_sublayers, prefix, attention,
                                                          edge_feat_nf):
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


# ---- inputs shared by tests/golden/make_golden.py (which runs the reference on them) and the tests ----------------
def ragged_fc_molecules(sizes, linkers, nf, seed):
    """Per-molecule dicts of a ragged fully-connected batch: fragments first, the linker atoms last."""
    g = torch.Generator().manual_seed(seed)
    mols = []
    for n, nl in zip(sizes, linkers):
        frag = torch.zeros(n)
        frag[:n - nl] = 1
        types = torch.randint(0, nf, (n,), generator=g)
        mols.append({'positions': 2.0 * torch.randn((n, 3), generator=g),
                     'one_hot': torch.nn.functional.one_hot(types, nf).float(),
                     'anchors': torch.zeros(n), 'fragment_mask': frag, 'linker_mask': 1 - frag,
                     'num_atoms': n, 'uuid': 0, 'name': 'm'})
    return mols


GLUE_HPARAMS = dict(in_node_nf=8, n_dims=3, context_node_nf=1, hidden_nf=128, activation='silu', tanh=False, n_layers=1,
                    attention=False, norm_constant=1e-6, inv_sublayers=2, sin_embedding=False, normalization_factor=100,
                    aggregation_method='sum', diffusion_steps=500, diffusion_noise_schedule='polynomial_2',
                    diffusion_noise_precision=1e-5, diffusion_loss_type='l2', normalize_factors=[1, 4, 10],
                    include_charges=False, model='egnn_dynamics', data_path='d', train_data_prefix='zinc_final_train',
                    val_data_prefix='zinc_final_val', batch_size=8, lr=2e-4, torch_device='cpu', test_epochs=20,
                    n_stability_samples=10, normalization='batch_norm', anchors_context=False)


def glue_cases():
    """(tag, hparam overrides, pockets, linker sizes asked of ``sample_fn``) of the ``ddpm_glue`` fixture."""
    return [
        ('fc', {}, False, [4, 2, 6]),
        ('fc_anchors', dict(anchors_context=True, context_node_nf=2, center_of_mass='anchors'), False, [3, 5, 2]),
        ('pocket', dict(train_data_prefix='MOAD_train.full', val_data_prefix='MOAD_val.full', context_node_nf=2,
                        graph_type='FC-10A-4A', in_node_nf=9), True, [3, 4]),
        ('pocket_anchors', dict(train_data_prefix='MOAD_train.full', val_data_prefix='MOAD_val.full', context_node_nf=3,
                                anchors_context=True, in_node_nf=9), True, [2, 5]),
    ]


def glue_molecules(pockets, nf, seed):
    """Per-molecule dicts as the reference's datasets / generation scripts build them (datasets.py:56-100,
    generate_with_pocket.py:233-248): fragments first, then (pockets) the pocket atoms, then the linker."""
    g = torch.Generator().manual_seed(seed)
    mols = []
    for n_frag, n_pocket, n_link in ([(9, 14, 4), (7, 11, 3)] if pockets else [(9, 0, 4), (6, 0, 2), (11, 0, 5)]):
        n = n_frag + n_pocket + n_link
        types = torch.randint(0, nf, (n,), generator=g)
        role = torch.cat([torch.zeros(n_frag), torch.ones(n_pocket), 2 * torch.ones(n_link)])
        anchors = torch.zeros(n)
        anchors[0] = 1
        anchors[n_frag - 1] = 1
        m = {'uuid': len(mols), 'name': f'm{len(mols)}', 'positions': 2.5 * torch.randn((n, 3), generator=g),
             'one_hot': torch.nn.functional.one_hot(types, nf).float(), 'anchors': anchors,
             'fragment_mask': (role < 2).float(), 'linker_mask': (role == 2).float(), 'num_atoms': n}
        if pockets:
            m['fragment_only_mask'] = (role == 0).float()
            m['pocket_mask'] = (role == 1).float()
        mols.append(m)
    return mols




FLAG_CASES = [                       # optional hyper-parameters of Dynamics (tests/golden/fc_forward_flags.npz)
    ('attention', dict(attention=True)),
    ('tanh', dict(tanh=True)),
    ('mean', dict(aggregation_method='mean')),
    ('all', dict(attention=True, tanh=True, aggregation_method='mean')),
    ('sin', dict(sin_embedding=True)),
]


def trained_like_state_dict(sd, seed, sigma=1.5, outlier=2.0 ** 10, bias_gain=30.0):
    """Seeded-random weights reshaped towards the statistics of a TRAINED checkpoint, the regime the a-priori power-of-two
    scales of the f16x3 arithmetic have to survive: the rows of the first layer of every edge / node /
    coordinate MLP are rescaled by a log-normal factor (``sigma``), ONE row of each by ``outlier`` on top, and every bias by
    ``bias_gain``; the columns of the layer that consumes those features are divided by the same factors, so the function
    stays of order one while the hidden activations, the row-L1 norms and the bias maxima the scales are derived from spread
    over ~20 binades.  The second edge layer's rows get their own log-normal factors (no outlier), undone in the columns of
    the node MLP that read the message sum / in the coordinate head's last layer."""
    rng = np.random.default_rng(seed)
    out = {k: v.clone().double() for k, v in sd.items()}
    for k in out:
        if k.endswith('.bias'):
            out[k] *= bias_gain

    def factors(n, with_outlier):
        f = np.exp(sigma * rng.standard_normal(n))
        if with_outlier:
            f[int(rng.integers(n))] *= outlier
        return torch.from_numpy(f)
    stems = sorted({k.rsplit('.', 2)[0] for k in out if k.endswith('_mlp.0.weight')})
    for stem in stems:                                     # e.g. dynamics.e_block_0.gcl_1.edge_mlp
        if stem.endswith('att_mlp'):
            continue
        f = factors(out[f'{stem}.0.weight'].shape[0], True)
        out[f'{stem}.0.weight'] *= f[:, None]
        out[f'{stem}.0.bias'] *= f
        out[f'{stem}.2.weight'] /= f[None, :]
        if stem.endswith('edge_mlp') or stem.endswith('coord_mlp'):
            g = factors(out[f'{stem}.2.weight'].shape[0], False)
            out[f'{stem}.2.weight'] *= g[:, None]
            out[f'{stem}.2.bias'] *= g
            if stem.endswith('coord_mlp'):
                out[f'{stem}.4.weight'] /= g[None, :]
            else:                                          # SiLU sits between: undone only in magnitude, which is the point
                node0 = stem.replace('edge_mlp', 'node_mlp') + '.0.weight'
                hid = out[node0].shape[1] // 2
                out[node0][:, hid:] /= g[None, :]
    return {k: v.to(sd[k].dtype) for k, v in out.items()}
