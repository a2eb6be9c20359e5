"""Oracle restatement of the EGNN denoiser (reference ``src/egnn.py``).

TEST INFRASTRUCTURE — see ``oracle/__init__.py``.  Plain PyTorch on CPU,
functional over a ``state_dict`` whose keys are the reference's
(``dynamics.embedding.weight``, ``dynamics.e_block_0.gcl_0.edge_mlp.0.weight``
...), in whatever dtype the tensors passed in have (fp32 for parity, fp64 for
a "truth" run).  The op ORDER follows the reference (edge list, gather, cat,
linear, SiLU, mask, scatter-add) so that fp32 results agree with the
unmodified reference to the last bit or two; pinned by tests/golden.

Scope: ``egnn_dynamics`` with SiLU; the optional hyper-parameters ``attention`` (egnn.py:42-43,52-54), ``tanh`` +
``coords_range`` (:85-99,104-105), ``aggregation_method='mean'`` (:315-319, counting masked edges too) and
``sin_embedding`` (:281-292) are restated as well (no released config uses them; pinned by
``tests/golden/fc_forward_flags.npz``).
"""
import math

from dataclasses import dataclass

import torch
import torch.nn.functional as F


@dataclass
class EGNNConfig:
    """Hyper-parameters of ``Dynamics.__init__`` (reference egnn.py:324-329)."""
    n_dims: int = 3
    in_node_nf: int = 9            # atom-type channels (nf)
    context_node_nf: int = 1
    hidden_nf: int = 128
    n_layers: int = 6
    inv_sublayers: int = 2
    norm_constant: float = 1e-6
    normalization_factor: float = 100.0
    condition_time: bool = True
    graph_type: str = 'FC'         # 'FC' | '4A' | 'FC-4A' | 'FC-10A-4A'
    centering: bool = False
    attention: bool = False
    tanh: bool = False
    coords_range: float = 15.0     # EGNN's default (egnn.py:183).  EGNN computes coords_range / n_layers (:192) but hands the
                                   # UNDIVIDED value to its blocks (:213), so every EquivariantUpdate uses 15
    aggregation_method: str = 'sum'
    sin_embedding: bool = False

    @property
    def fin(self):
        return self.in_node_nf + self.context_node_nf + int(self.condition_time)


def fc_edges(n_nodes, batch_size, device=None):
    """Fully-connected edge list, self-loops included, edge e = b*N*N + i*N + j.

    Reference: ``Dynamics.get_edges`` egnn.py:449-467 (python triple loop);
    here by index arithmetic, same order.
    """
    e = torch.arange(batch_size * n_nodes * n_nodes, device=device)
    b = e // (n_nodes * n_nodes)
    i = (e // n_nodes) % n_nodes
    j = e % n_nodes
    return b * n_nodes + i, b * n_nodes + j


def coord2diff(x, row, col, norm_constant=1.0):
    """Squared distance and normalised difference per edge (egnn.py:295-301)."""
    diff = x[row] - x[col]
    radial = torch.sum(diff ** 2, 1).unsqueeze(1)
    norm = torch.sqrt(radial + 1e-8)
    return radial, diff / (norm + norm_constant)


def segment_sum(data, row, num_segments, normalization_factor, aggregation_method='sum'):
    """``unsorted_segment_sum`` (egnn.py:304-320): 'sum' divides by the normalisation factor, 'mean' by the number of
    edges of the segment — every edge of the list counts, masked ones included (the count is taken on ``ones``)."""
    out = data.new_zeros((num_segments, data.size(1)))
    idx = row.unsqueeze(-1).expand(-1, data.size(1))
    out.scatter_add_(0, idx, data)
    if aggregation_method == 'sum':
        return out / normalization_factor
    assert aggregation_method == 'mean'
    norm = data.new_zeros(out.shape)
    norm.scatter_add_(0, idx, data.new_ones(data.shape))
    norm[norm == 0] = 1
    return out / norm


def sin_embedding(x, max_res=15., min_res=15. / 2000., div_factor=4):
    """``SinusoidsEmbeddingNew`` (egnn.py:281-292) of squared distances ``[E,1]`` -> ``[E,12]``."""
    n_freq = int(math.log(max_res / min_res, div_factor)) + 1
    freq = (2 * math.pi * div_factor ** torch.arange(n_freq) / max_res).to(x.dtype)
    emb = torch.sqrt(x + 1e-8) * freq[None, :]
    return torch.cat((emb.sin(), emb.cos()), dim=-1)


def _lin(p, key, x):
    return F.linear(x, p[key + '.weight'], p.get(key + '.bias'))


def gcl(p, pre, h, row, col, edge_attr, node_mask, edge_mask, cfg):
    """One GCL: edge MLP -> mask -> sum_j /norm -> node MLP + residual -> mask.

    Reference: ``GCL.forward/edge_model/node_model`` egnn.py:45-80, MLPs :19-30.
    """
    inp = torch.cat([h[row], h[col], edge_attr], dim=1)
    m = F.silu(_lin(p, pre + '.edge_mlp.0', inp))
    m = F.silu(_lin(p, pre + '.edge_mlp.2', m))
    if cfg.attention:                                             # egnn.py:52-54
        m = m * torch.sigmoid(_lin(p, pre + '.att_mlp.0', m))
    if edge_mask is not None:
        m = m * edge_mask
    agg = segment_sum(m, row, h.size(0), cfg.normalization_factor, cfg.aggregation_method)
    t = torch.cat([h, agg], dim=1)
    t = F.silu(_lin(p, pre + '.node_mlp.0', t))
    h = h + _lin(p, pre + '.node_mlp.2', t)
    if node_mask is not None:
        h = h * node_mask
    return h


def equivariant_update(p, pre, h, x, row, col, coord_diff, edge_attr, linker_mask, node_mask, edge_mask, cfg):
    """Coordinate update (egnn.py:101-125, MLP :90-97; last layer has no bias)."""
    inp = torch.cat([h[row], h[col], edge_attr], dim=1)
    s = F.silu(_lin(p, pre + '.coord_mlp.0', inp))
    s = F.silu(_lin(p, pre + '.coord_mlp.2', s))
    s = _lin(p, pre + '.coord_mlp.4', s)
    if cfg.tanh:                                                  # egnn.py:104-105
        trans = coord_diff * torch.tanh(s) * float(cfg.coords_range)
    else:
        trans = coord_diff * s
    if edge_mask is not None:
        trans = trans * edge_mask
    agg = segment_sum(trans, row, x.size(0), cfg.normalization_factor, cfg.aggregation_method)
    if linker_mask is not None:
        agg = agg * linker_mask
    x = x + agg
    if node_mask is not None:
        x = x * node_mask
    return x


def equivariant_block(p, pre, h, x, row, col, d0, node_mask, linker_mask, edge_mask, cfg):
    """``EquivariantBlock.forward`` egnn.py:157-178."""
    radial, coord_diff = coord2diff(x, row, col, cfg.norm_constant)
    if cfg.sin_embedding:                                         # egnn.py:160-161
        radial = sin_embedding(radial)
    edge_attr = torch.cat([radial, d0], dim=1)
    for i in range(cfg.inv_sublayers):
        h = gcl(p, f'{pre}.gcl_{i}', h, row, col, edge_attr, node_mask, edge_mask, cfg)
    x = equivariant_update(p, f'{pre}.gcl_equiv', h, x, row, col, coord_diff, edge_attr,
                           linker_mask, node_mask, edge_mask, cfg)
    if node_mask is not None:
        h = h * node_mask
    return h, x


def egnn_forward(p, pre, h, x, row, col, node_mask, linker_mask, edge_mask, cfg):
    """``EGNN.forward`` egnn.py:218-238 (d0 uses coord2diff's default norm, radial only)."""
    d0, _ = coord2diff(x, row, col)
    if cfg.sin_embedding:                                         # egnn.py:221-222
        d0 = sin_embedding(d0)
    h = _lin(p, pre + '.embedding', h)
    for i in range(cfg.n_layers):
        h, x = equivariant_block(p, f'{pre}.e_block_{i}', h, x, row, col, d0,
                                 node_mask, linker_mask, edge_mask, cfg)
    h = _lin(p, pre + '.embedding_out', h)
    if node_mask is not None:
        h = h * node_mask
    return h, x


def find_nan_idx(z):
    """Per-sample NaN index set (``FoundNaNException.find_nan_idx`` utils.py:283-289)."""
    return {i for i in range(z.shape[0]) if bool(torch.any(torch.isnan(z[i])))}


class OracleNaN(Exception):
    """Stand-in for ``utils.FoundNaNException`` (utils.py:274-289)."""

    def __init__(self, x, h):
        xs, hs = find_nan_idx(x), find_nan_idx(h)
        self.x_h_nan_idx = xs & hs
        self.only_x_nan_idx = xs - hs
        self.only_h_nan_idx = hs - xs


def _node_inputs(cfg, t, xh, node_mask, context):
    """Flatten, mask, append time + context (egnn.py:385-407 / :480-512)."""
    bs, n = xh.shape[0], xh.shape[1]
    nm = node_mask.view(bs * n, 1)
    xh = xh.reshape(bs * n, -1).clone() * nm
    x = xh[:, :cfg.n_dims].clone()
    h = xh[:, cfg.n_dims:].clone()
    if cfg.condition_time:
        if t.numel() == 1:
            h_time = torch.empty_like(h[:, 0:1]).fill_(t.item())
        else:
            h_time = t.view(bs, 1).repeat(1, n).view(bs * n, 1).to(h.dtype)
        h = torch.cat([h, h_time], dim=1)
    if context is not None:
        h = torch.cat([h, context.reshape(bs * n, cfg.context_node_nf).to(h.dtype)], dim=1)
    return x, h, nm


def _finish(cfg, bs, n, x, x_final, h_final, nm, context):
    """Velocity, strip context/time, NaN check (egnn.py:420-447)."""
    vel = (x_final - x) * nm
    if context is not None:
        h_final = h_final[:, :-cfg.context_node_nf]
    if cfg.condition_time:
        h_final = h_final[:, :-1]
    vel = vel.view(bs, n, -1)
    h_final = h_final.view(bs, n, -1)
    if torch.any(torch.isnan(vel)) or torch.any(torch.isnan(h_final)):
        raise OracleNaN(vel, h_final)
    if cfg.centering:
        nm3 = nm.view(bs, n, 1)
        cnt = nm3.sum(1, keepdims=True)
        vel = vel - (vel.sum(1, keepdim=True) / cnt) * nm3      # utils.py:56-63
    return torch.cat([vel, h_final], dim=2)


def dynamics_forward(p, cfg, t, xh, node_mask, linker_mask, edge_mask, context, pre='dynamics'):
    """``Dynamics.forward`` (fully-connected graph) egnn.py:374-447.

    ``p`` holds the ``Dynamics`` module's state_dict (keys ``dynamics.*``);
    ``edge_mask`` is the int8 ``[B*N*N,1]`` tensor of ``collate`` with values
    {0,-1,-2} (datasets.py:366-369) — it multiplies every message as-is.
    """
    assert cfg.graph_type == 'FC'
    bs, n = xh.shape[0], xh.shape[1]
    row, col = fc_edges(n, bs, xh.device)
    x, h, nm = _node_inputs(cfg, t, xh, node_mask, context)
    lm = linker_mask.view(bs * n, 1) if linker_mask is not None else None
    h_final, x_final = egnn_forward(p, pre, h, x, row, col, nm, lm, edge_mask, cfg)
    return _finish(cfg, bs, n, x, x_final, h_final, nm, context)


def pocket_edges(cfg, x, node_mask, batch_mask, linker_mask, fragment_only_mask, pocket_only_mask):
    """Radius-graph edge list of ``DynamicsWithPockets`` (egnn.py:554-596).

    Returns (row, col) sorted by (row, col) like ``torch.where`` does.
    """
    nm = node_mask.squeeze(-1).bool()
    same_graph = batch_mask[:, None] == batch_mask[None, :]
    both_real = nm[:, None] & nm[None, :]
    no_self = ~torch.eye(x.size(0), dtype=torch.bool, device=x.device)
    base = same_graph & both_real & no_self
    dist = torch.cdist(x, x)
    if cfg.graph_type == '4A':                                   # egnn.py:554-563
        adj = base & (dist <= 4)
    else:
        lig = (linker_mask.squeeze(-1).bool() & nm) | (fragment_only_mask.squeeze(-1).bool() & nm)
        poc = pocket_only_mask.squeeze(-1).bool() & nm
        cut = 4 if cfg.graph_type == 'FC-4A' else 10
        lig_lig = lig[:, None] & lig[None, :]
        poc_poc = (poc[:, None] & poc[None, :]) & (dist <= 4)
        cross = ((lig[:, None] & poc[None, :]) | (poc[:, None] & lig[None, :])) & (dist <= cut)
        adj = (lig_lig | poc_poc | cross) & base
    row, col = torch.where(adj)
    return row, col


def dynamics_forward_pockets(p, cfg, t, xh, node_mask, linker_mask, edge_mask, context, pre='dynamics'):
    """``DynamicsWithPockets.forward`` egnn.py:471-552.

    ``edge_mask`` is the per-node batch-index vector ``[B*N]`` (datasets.py:359-364);
    the last two context channels are the fragment-only / pocket-only masks; the
    EGNN runs with ``edge_mask=None`` (no sign flip, no diagonal; egnn.py:517-524).
    """
    assert cfg.graph_type in ('4A', 'FC-4A', 'FC-10A-4A')
    bs, n = xh.shape[0], xh.shape[1]
    lm = linker_mask.view(bs * n, 1)
    frag_only = context[..., -2].reshape(bs * n, 1)
    pock_only = context[..., -1].reshape(bs * n, 1)
    x, h, nm = _node_inputs(cfg, t, xh, node_mask, context)
    assert torch.all((frag_only.bool() | pock_only.bool() | lm.bool()) == nm.bool())
    row, col = pocket_edges(cfg, x, nm, edge_mask.view(-1), lm, frag_only, pock_only)
    h_final, x_final = egnn_forward(p, pre, h, x, row, col, nm, lm, None, cfg)
    return _finish(cfg, bs, n, x, x_final, h_final, nm, context)


def flops_min(cfg, pairs, nodes):
    """Algorithmic FLOPs per forward, SURVEY.md section 8(d) ``F_min``."""
    hn = cfg.hidden_nf
    per_pair = 2 * (3 * (hn * hn + 2 * hn) + hn)
    per_node = 2 * (2 * (2 * hn * hn + hn * hn)) + 2 * 3 * (2 * hn * hn)
    return cfg.n_layers * (pairs * per_pair + nodes * per_node) + nodes * 4 * cfg.fin * hn
