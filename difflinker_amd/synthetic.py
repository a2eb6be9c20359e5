"""Seeded synthetic fragment/linker batches of the shapes BASELINE.json names.

There are no datasets or checkpoints offline, so parity tests and ``bench.py`` run on
synthetic molecules shaped like the reference's batches: per-molecule
dicts with the keys of ``src/datasets.py`` (``positions``, ``one_hot``, ``anchors``,
``fragment_mask``, ``linker_mask`` [, ``pocket_mask``, ``fragment_only_mask``]) pushed through
the same ``collate`` the sampler uses, so the int8 ``{0,-1,-2}`` edge mask / pocket batch-id
vector are produced by the code path a real batch takes.
"""
import math

import torch

from . import const
from .datasets import collate

# name -> (nf, ctx, n_layers, batch, N, (n_lo, n_hi), (link_lo, link_hi), T, graph_type)
CONFIGS = {
    'C1': dict(nf=8, ctx=1, n_layers=8, batch=8, n_max=30, n_lo=24, linker=(3, 8), T=50, graph_type='FC'),
    'C2': dict(nf=9, ctx=1, n_layers=6, batch=256, n_max=50, n_lo=35, linker=(3, 12), T=500, graph_type='FC'),
    'C4': dict(nf=9, ctx=2, n_layers=6, batch=64, n_frag=30, n_pocket=250, linker=(6, 12), T=500,
               graph_type='FC-10A-4A'),
    # BASELINE config 5, per-GPU shard: the C4 molecules, 64 per GPU (512 over 8), an EDM built with timesteps = 1000 and
    # sampled over all of them (no duplicate-gamma steps)
    'C5': dict(nf=9, ctx=2, n_layers=6, batch=64, n_frag=30, n_pocket=250, linker=(6, 12), T=1000, timesteps=1000,
               graph_type='FC-10A-4A'),
    # not a BASELINE configuration: GEOM hparams on molecules of 60..80 atoms - beyond one compute unit's LDS (55 atoms):
    # round 3 runs them fused on teams of two compute units, round 2 on the HBM-resident kernels under a host-driven loop
    'C2L': dict(nf=9, ctx=1, n_layers=6, batch=64, n_max=80, n_lo=60, linker=(3, 12), T=500, graph_type='FC'),
    # beyond the fused paths (> 110 atoms per molecule): HBM-resident per-pass kernels under a host-driven T-step loop
    'C2XL': dict(nf=9, ctx=1, n_layers=6, batch=32, n_max=150, n_lo=120, linker=(3, 12), T=500, graph_type='FC'),
}


def _randint(g, lo, hi):
    return int(torch.randint(lo, hi + 1, (1,), generator=g))


def _ball(g, n, radius):
    """n points uniform in a ball of the given radius."""
    d = torch.randn((n, 3), generator=g)
    d = d / d.norm(dim=1, keepdim=True).clamp_min(1e-12)
    r = radius * torch.rand((n, 1), generator=g) ** (1.0 / 3.0)
    return d * r


def fc_molecules(batch, n_max, n_lo, linker, nf, seed, uniform_size=False):
    """List of per-molecule dicts for the fully-connected (ZINC / GEOM) configs.

    n_b ~ U{n_lo..n_max} (molecule 0 has n_max atoms so the padded N is n_max), linker size
    ~ U{linker} taken from the LAST rows, positions N(0, 2.5 A), atom types uniform, one anchor
    per fragment side.  ``uniform_size`` makes every molecule n_max atoms (the unpadded variant).
    """
    g = torch.Generator().manual_seed(seed)
    mols = []
    for b in range(batch):
        n = n_max if (b == 0 or uniform_size) else _randint(g, n_lo, n_max)
        n_link = _randint(g, linker[0], linker[1])
        n_frag = n - n_link
        pos = 2.5 * torch.randn((n, 3), generator=g)
        types = torch.randint(0, nf, (n,), generator=g)
        one_hot = torch.nn.functional.one_hot(types, nf).to(const.TORCH_FLOAT)
        frag = torch.zeros(n)
        frag[:n_frag] = 1
        link = 1 - frag
        anchors = torch.zeros(n)
        anchors[0] = 1
        anchors[n_frag - 1] = 1
        mols.append({
            'uuid': b, 'name': f'synthetic_{b}', 'num_atoms': n,
            'positions': pos.to(const.TORCH_FLOAT), 'one_hot': one_hot, 'anchors': anchors,
            'fragment_mask': frag, 'linker_mask': link,
        })
    return mols


def pocket_molecules(batch, n_frag, n_pocket, linker, nf, seed, min_gap=1e-4):
    """Per-molecule dicts for the pocket-conditioned config (C4): ``n_frag`` fragment atoms
    uniform in a 5 A ball, ``n_pocket`` pocket atoms uniform in a 10 A ball, then the linker
    rows.  ``fragment_mask`` covers fragment+pocket atoms like the MOAD data
    (lightning.py:431-433).  Pairs closer than ``min_gap`` to the 4 A / 10 A cut-offs are
    re-drawn so edge membership does not flip between implementations."""
    g = torch.Generator().manual_seed(seed)
    mols = []
    for b in range(batch):
        n_link = _randint(g, linker[0], linker[1])
        fixed = torch.cat([_ball(g, n_frag, 5.0), _ball(g, n_pocket, 10.0)])
        radius = torch.cat([torch.full((n_frag,), 5.0), torch.full((n_pocket,), 10.0)])
        for _ in range(1000):                       # re-draw only the atoms of pairs that sit on a cut-off
            d = torch.cdist(fixed.double(), fixed.double())
            bad = ((d - 4.0).abs() < min_gap) | ((d - 10.0).abs() < min_gap)
            rows = torch.nonzero(bad.any(dim=1)).flatten()
            if rows.numel() == 0:
                break
            for a in rows[::2].tolist() or rows.tolist():
                fixed[a] = _ball(g, 1, float(radius[a]))[0]
        link_pos = 2.0 * torch.randn((n_link, 3), generator=g)
        n = n_frag + n_pocket + n_link
        pos = torch.cat([fixed, link_pos]).to(const.TORCH_FLOAT)
        types = torch.randint(0, nf, (n,), generator=g)
        one_hot = torch.nn.functional.one_hot(types, nf).to(const.TORCH_FLOAT)
        frag_only = torch.zeros(n)
        frag_only[:n_frag] = 1
        pocket = torch.zeros(n)
        pocket[n_frag:n_frag + n_pocket] = 1
        link = torch.zeros(n)
        link[n_frag + n_pocket:] = 1
        anchors = torch.zeros(n)
        anchors[0] = 1
        anchors[n_frag - 1] = 1
        mols.append({
            'uuid': b, 'name': f'synthetic_pocket_{b}', 'num_atoms': n,
            'positions': pos, 'one_hot': one_hot, 'anchors': anchors,
            'fragment_mask': frag_only + pocket, 'linker_mask': link,
            'fragment_only_mask': frag_only, 'pocket_mask': pocket,
        })
    return mols


def make_batch(name, seed=0, batch=None, uniform_size=False, device='cpu'):
    """Collated batch dict for config ``name`` ('C1' | 'C2' | 'C4'), plus its config dict."""
    cfg = dict(CONFIGS[name])
    if batch is not None:
        cfg['batch'] = batch
    if cfg['graph_type'] == 'FC':
        mols = fc_molecules(cfg['batch'], cfg['n_max'], cfg['n_lo'], cfg['linker'], cfg['nf'], seed, uniform_size)
    else:
        mols = pocket_molecules(cfg['batch'], cfg['n_frag'], cfg['n_pocket'], cfg['linker'], cfg['nf'], seed)
    data = collate(mols)
    data = {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in data.items()}
    return data, cfg


def sampler_inputs(data, pockets=False, anchors_context=False):
    """The tensors ``DDPM.sample_chain`` hands to ``EDM.sample_chain`` for a collated batch
    whose linker rows are already in place (lightning.py:413-451): context assembly and
    fragment-COM removal."""
    x, h = data['positions'], data['one_hot']
    node_mask, edge_mask = data['atom_mask'], data['edge_mask']
    fragment_mask, linker_mask = data['fragment_mask'], data['linker_mask']
    if pockets:
        frag_only = data['fragment_only_mask']
        pocket_only = fragment_mask - frag_only
        parts = [frag_only, pocket_only]
        com_mask = frag_only
    else:
        parts = [fragment_mask]
        com_mask = fragment_mask
    if anchors_context:
        parts = [data['anchors']] + parts
    context = torch.cat(parts, dim=-1)
    # linker rows start from the zero template (datasets.py:476-480) ...
    x = x * fragment_mask
    h = h * fragment_mask
    # ... then the fragment centre of mass is removed from every real atom (utils.py:66-74)
    n = com_mask.sum(1, keepdims=True)
    mean = torch.sum(x * com_mask, dim=1, keepdim=True) / n
    x = x - mean * node_mask
    return dict(x=x, h=h, node_mask=node_mask, fragment_mask=fragment_mask, linker_mask=linker_mask,
                edge_mask=edge_mask, context=context)


def pair_and_node_counts(data, pockets=False):
    """(P, V) of SURVEY 8d: FC pairs sum n_b^2 (diagonal included) and real atoms."""
    n_b = data['atom_mask'].view(data['atom_mask'].shape[0], -1).sum(1).to(torch.int64)
    return int((n_b * n_b).sum()), int(n_b.sum())


def flops_min(hidden_nf, n_layers, fin, pairs, nodes):
    """Algorithmic FLOPs per ``Dynamics.forward`` (``F_min``)."""
    hn = hidden_nf
    per_pair = 2 * (3 * (hn * hn + 2 * hn) + hn)
    per_node = 2 * (2 * (2 * hn * hn + hn * hn)) + 2 * 3 * (2 * hn * hn)
    return n_layers * (pairs * per_pair + nodes * per_node) + nodes * 4 * fin * hn


def flops_executed(hidden_nf, n_layers, fin, pairs, pairs_coord, nodes):
    """FLOPs the kernels really spend per ``Dynamics.forward``: ``flops_min`` with the coordinate head's edge model counted
    only on the ``pairs_coord`` pairs whose receiving atom is inside the linker mask (the reference multiplies every other
    atom's coordinate sum by zero, egnn.py:113-116, and the kernels do not compute what it discards)."""
    hn = hidden_nf
    gcl_pair = 2 * (2 * (hn * hn + 2 * hn))
    coord_pair = 2 * ((hn * hn + 2 * hn) + hn)
    per_node = 2 * (2 * (2 * hn * hn + hn * hn)) + 2 * 3 * (2 * hn * hn)
    return n_layers * (pairs * gcl_pair + pairs_coord * coord_pair + nodes * per_node) + nodes * 4 * fin * hn


def split_terms(precision, hidden_nf, n_layers, fin, pairs, pairs_coord, nodes):
    """fp16 MFMA terms per multiply-accumulate, averaged over the executed work: 3 in 'f16x3' (hi*hi' + hi*lo' + lo*hi'), and in
    'f16x2' 2 for the GCL edge models' second layer (the activation enters as one fp16) and 3 for everything else.  The
    algorithmic-flop ceiling of a split scheme is the dense f16 MFMA peak divided by this figure."""
    if precision != 'f16x2':
        return 3.0
    hn = hidden_nf
    total = flops_executed(hidden_nf, n_layers, fin, pairs, pairs_coord, nodes)
    gcl = n_layers * pairs * 2 * (2 * (hn * hn + 2 * hn))
    return (2.0 * gcl + 3.0 * (total - gcl)) / total


def coord_pair_count(data):
    """FC graphs: pairs (i, j) of a molecule whose receiving atom i is a linker atom = sum_b n_linker_b * n_b."""
    n_b = data['atom_mask'].view(data['atom_mask'].shape[0], -1).sum(1).to(torch.int64)
    n_l = data['linker_mask'].view(data['linker_mask'].shape[0], -1).sum(1).to(torch.int64)
    return int((n_l * n_b).sum())


def layer_bytes(nodes, pairs_fc):
    """Algorithmic HBM bytes per EquivariantBlock with h entering/leaving once (SURVEY 8d ``A_layer``)."""
    return 2 * nodes * 128 * 4 + 3 * nodes * 12 + nodes * 2 + pairs_fc
