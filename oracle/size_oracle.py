"""TEST INFRASTRUCTURE ONLY — CPU restatement of the linker-size predictor.

Functional PyTorch-CPU restatement of ``SizeGNN.forward`` (reference ``src/linker_size.py:45-91``) and of the
inference part of ``SizeClassifier.forward`` (``src/linker_size_lightning.py:83-117``): explicit edge list,
gather -> cat -> Linear -> ReLU -> mask -> scatter-add, exactly the reference's op sequence.  Pinned to outputs of the
unmodified reference modules by ``tests/golden/size_gnn.npz`` (``tests/golden/make_golden.py``).

Only ``tests/`` may import this module; the product path (``difflinker_amd.linker_size``) runs in HIP and never
touches it.
"""
import torch
import torch.nn.functional as F

from .egnn_oracle import coord2diff, fc_edges, segment_sum


def _lin(p, key, x):
    return F.linear(x, p[key + '.weight'], p.get(key + '.bias'))


def _bn_eval(p, key, x, eps=1e-5):
    """nn.BatchNorm1d in eval mode (running statistics), egnn.py:31-38."""
    return (x - p[key + '.running_mean']) / torch.sqrt(p[key + '.running_var'] + eps) * p[key + '.weight'] + p[key + '.bias']


def size_gcl(p, pre, h, row, col, edge_attr, node_mask, edge_mask, batch_norm=False):
    """GCL with ReLU, one edge attribute, normalization_factor = 1, 'sum' (linker_size.py:53-63; egnn.py:45-80)."""
    m = F.relu(_lin(p, pre + '.edge_mlp.0', torch.cat([h[row], h[col], edge_attr], dim=1)))
    m = F.relu(_lin(p, pre + '.edge_mlp.2', m))
    m = m * edge_mask
    agg = segment_sum(m, row, h.size(0), 1.0)
    t = torch.cat([h, agg], dim=1)
    if not batch_norm:
        t = F.relu(_lin(p, pre + '.node_mlp.0', t))
        out = _lin(p, pre + '.node_mlp.2', t)
    else:
        t = F.relu(_bn_eval(p, pre + '.node_mlp.1', _lin(p, pre + '.node_mlp.0', t)))
        out = _bn_eval(p, pre + '.node_mlp.4', _lin(p, pre + '.node_mlp.3', t))
    return (h + out) * node_mask


def size_gnn_forward(p, h, row, col, distances, node_mask, edge_mask, n_layers, batch_norm=False, pre=''):
    """``SizeGNN.forward`` (linker_size.py:83-91)."""
    h = _lin(p, pre + 'embedding_in', h)
    h = size_gcl(p, pre + 'gcl1', h, row, col, distances, node_mask, edge_mask, batch_norm)
    for i in range(n_layers - 1):
        h = size_gcl(p, pre + f'gcl_layers.{i}', h, row, col, distances, node_mask, edge_mask, batch_norm)
    return _lin(p, pre + 'embedding_out', h)


def size_classifier_logits(p, one_hot, positions, fragment_mask, edge_mask, n_layers, batch_norm=False, pre='gnn.'):
    """Inference path of ``SizeClassifier.forward`` (linker_size_lightning.py:83-110): fragments only, squared
    distances as the edge attribute, edges kept where ``edge_mask & (squared distance < 6)``, mean over ALL padded
    nodes."""
    bs, n = positions.shape[:2]
    x = (positions * fragment_mask).reshape(bs * n, -1)
    h = (one_hot * fragment_mask).reshape(bs * n, -1)
    row, col = fc_edges(n, bs)
    distances, _ = coord2diff(x, row, col)
    dmask = (edge_mask.reshape(-1, 1).bool() & (distances < 6)).to(h.dtype)
    out = size_gnn_forward(p, h, row, col, distances, fragment_mask.reshape(bs * n, 1), dmask, n_layers, batch_norm, pre)
    return out.view(bs, n, -1).mean(1)
