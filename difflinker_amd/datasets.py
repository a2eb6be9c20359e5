"""Batch assembly for the sampling boundary (reference ``src/datasets.py:332-375, 476-512``).

Only the two functions ``DDPM.sample_chain`` needs: ``collate`` (padding, int8 atom/edge masks)
and ``create_templates_for_linker_generation``.  The dataset classes / SDF preprocessing of the
reference are RDKit-bound I/O and out of scope.

The int8 quirk is part of the contract: ``edge_mask`` is built in int8 and multiplied by
``~eye`` (bitwise NOT on int8: 0 -> -1, 1 -> -2), so real i!=j pairs carry -1, real self
pairs -2 and any pair with a padded endpoint 0 (datasets.py:366-369, const.py:7).
"""
import os

import torch

from . import const


class ZincDataset(torch.utils.data.Dataset):
    """List of per-molecule dicts from ``<data_path>/<prefix>.pt`` (datasets.py:40-54).  Building that file from the
    raw SDF tables is RDKit preprocessing (datasets.py:56-100) and out of scope: a missing file raises."""

    def __init__(self, data_path, prefix, device):
        dataset_path = os.path.join(data_path, f'{prefix}.pt')
        if not os.path.exists(dataset_path):
            raise FileNotFoundError(f'{dataset_path} not found: preprocess the dataset with the reference tooling '
                                    '(RDKit) first')
        self.data = torch.load(dataset_path, map_location=device, weights_only=False)

    def __len__(self):
        return len(self.data)

    def __getitem__(self, item):
        return self.data[item]


class MOADDataset(torch.utils.data.Dataset):
    """Pocket-conditioned dataset: in-memory ``data`` or ``<data_path>/<prefix>_<pocket_mode>.pt`` for prefixes like
    ``MOAD_test.full`` / ``MOAD_test_full`` (datasets.py:103-129)."""

    def __init__(self, data=None, data_path=None, prefix=None, device=None):
        assert (data is not None) or all(x is not None for x in (data_path, prefix, device))
        if data is not None:
            self.data = data
            return
        if '.' in prefix:
            prefix, pocket_mode = prefix.split('.')
        else:
            parts = prefix.split('_')
            prefix, pocket_mode = '_'.join(parts[:-1]), parts[-1]
        dataset_path = os.path.join(data_path, f'{prefix}_{pocket_mode}.pt')
        if not os.path.exists(dataset_path):
            raise FileNotFoundError(f'{dataset_path} not found: preprocess the dataset with the reference tooling '
                                    '(RDKit) first')
        self.data = torch.load(dataset_path, map_location=device, weights_only=False)

    def __len__(self):
        return len(self.data)

    def __getitem__(self, item):
        return self.data[item]


def collate(batch):
    """Pad a list of per-molecule dicts into ``[B,N,...]`` tensors (datasets.py:332-375)."""
    out = {}
    for data in batch:
        for key, value in data.items():
            out.setdefault(key, []).append(value)

    for key, value in out.items():
        if key in const.DATA_LIST_ATTRS:
            continue
        if key in const.DATA_ATTRS_TO_PAD:
            out[key] = torch.nn.utils.rnn.pad_sequence(value, batch_first=True, padding_value=0)
            continue
        raise Exception(f'Unknown batch key: {key}')

    atom_mask = (out['fragment_mask'].bool() | out['linker_mask'].bool()).to(const.TORCH_INT)
    out['atom_mask'] = atom_mask[:, :, None]
    batch_size, n_nodes = atom_mask.size()

    if 'pocket_mask' in batch[0].keys():
        # MOAD / pockets: "edge_mask" is the per-node batch index (datasets.py:359-364)
        out['edge_mask'] = torch.arange(batch_size, dtype=const.TORCH_INT, device=atom_mask.device) \
            .repeat_interleave(n_nodes)
    else:
        edge_mask = atom_mask[:, None, :] * atom_mask[:, :, None]
        diag_mask = ~torch.eye(n_nodes, dtype=const.TORCH_INT, device=atom_mask.device).unsqueeze(0)
        edge_mask = edge_mask * diag_mask                      # {0,-1,-2}
        out['edge_mask'] = edge_mask.view(batch_size * n_nodes * n_nodes, 1)

    for key in const.DATA_ATTRS_TO_ADD_LAST_DIM:
        if key in out.keys():
            out[key] = out[key][:, :, None]
    return out


def collate_with_fragment_without_pocket_edges(batch):
    """``collate_with_fragment_edges`` for pocket-conditioned inputs (datasets.py:425-469): the edge mask covers the
    FRAGMENT atoms only ('fragment_only_mask'); pocket atoms stay nodes of the batch but get no size-predictor edges."""
    return collate_with_fragment_edges(batch, mask_key='fragment_only_mask')


def collate_with_fragment_edges(batch, mask_key='fragment_mask'):
    """``collate`` variant used by the generation scripts and the size predictor (datasets.py:378-422): the edge mask
    covers FRAGMENT atoms only and an explicit fully-connected edge list ``[rows, cols]`` (e = b*N*N + i*N + j) is
    attached.  The mask is the float product ``frag_i * frag_j * ~eye`` with ``~eye`` taken on int8, i.e. -1 off the
    diagonal and -2 ON it (datasets.py:396-399): consumers that call ``.bool()`` on it keep the self loops."""
    out = {}
    for data in batch:
        for key, value in data.items():
            out.setdefault(key, []).append(value)

    for key, value in out.items():
        if key in const.DATA_LIST_ATTRS:
            continue
        if key in const.DATA_ATTRS_TO_PAD:
            out[key] = torch.nn.utils.rnn.pad_sequence(value, batch_first=True, padding_value=0)
            continue
        raise Exception(f'Unknown batch key: {key}')

    frag_mask = out[mask_key]
    batch_size, n_nodes = frag_mask.size()
    diag_mask = ~torch.eye(n_nodes, dtype=const.TORCH_INT, device=frag_mask.device).unsqueeze(0)
    edge_mask = frag_mask[:, None, :] * frag_mask[:, :, None] * diag_mask
    out['edge_mask'] = edge_mask.view(batch_size * n_nodes * n_nodes, 1)

    idx = torch.arange(batch_size * n_nodes * n_nodes, device=frag_mask.device)
    base = (idx // (n_nodes * n_nodes)) * n_nodes
    out['edges'] = [base + (idx // n_nodes) % n_nodes, base + idx % n_nodes]

    atom_mask = (out['fragment_mask'].bool() | out['linker_mask'].bool()).to(const.TORCH_INT)
    out['atom_mask'] = atom_mask[:, :, None]
    for key in const.DATA_ATTRS_TO_ADD_LAST_DIM:
        if key in out.keys():
            out[key] = out[key][:, :, None]
    return out


def create_template(tensor, fragment_size, linker_size, fill=0):
    """Keep the fragment rows, append ``linker_size`` constant rows (datasets.py:476-480)."""
    keep = tensor[:fragment_size]
    add = torch.full((int(linker_size), tensor.shape[1]), fill, dtype=keep.dtype, device=keep.device)
    return torch.cat([keep, add], dim=0)


def create_templates_for_linker_generation(data, linker_sizes):
    """Replace every molecule's linker rows by a zero template of the requested size
    and re-collate (datasets.py:483-512)."""
    decoupled = []
    for i, linker_size in enumerate(linker_sizes):
        item = {}
        fragment_size = data['fragment_mask'][i].squeeze().sum().int()
        for k, v in data.items():
            if k == 'num_atoms':
                item[k] = fragment_size + linker_size
            elif k in const.DATA_LIST_ATTRS:
                item[k] = v[i]
            elif k in const.DATA_ATTRS_TO_PAD:
                t = create_template(v[i], fragment_size, linker_size, fill=1 if k == 'linker_mask' else 0)
                if k in const.DATA_ATTRS_TO_ADD_LAST_DIM:
                    t = t.squeeze(-1)
                item[k] = t
        decoupled.append(item)
    return collate(decoupled)


def get_dataloader(dataset, batch_size, collate_fn=collate, shuffle=False):
    """datasets.py:472-473."""
    return torch.utils.data.DataLoader(dataset, batch_size, collate_fn=collate_fn, shuffle=shuffle)
