"""Generation drivers — the ``main()`` bodies of the reference's ``generate.py`` (:62-167),
``generate_with_pocket.py`` (:116-283) and ``generate_with_protein.py`` (:151-300) as importable functions on top of
``DDPM.sample_chain`` (HIP), ``SizeClassifier`` (HIP) and the RDKit-free I/O of ``io.py``.

Differences from the scripts, all outside the sampling path: the ``obabel xyz -> sdf`` conversion is not run (no
OpenBabel here; the ``.xyz`` files are what the reference's own post-processing starts from), and the functions
return the list of written files instead of printing.  ``python -m difflinker_amd.generate --help`` exposes the same
flags as the three scripts (``--pocket`` / ``--protein`` select the pocket-conditioned variants).
"""
import argparse
import os
import random

import numpy as np
import torch

from . import const
from .datasets import MOADDataset, collate_with_fragment_edges, collate_with_fragment_without_pocket_edges
from .io import get_pocket, parse_molecule, pocket_arrays, read_molecule, read_pocket, save_xyz_file
from .lightning import DDPM
from .linker_size import SizeClassifier
from .utils import FoundNaNException


def set_deterministic(seed):
    """src/utils.py:263-271."""
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def make_sample_fn(linker_size, device, with_pocket=False):
    """``linker_size``: an integer, ``"lo,hi"`` bounds (uniform, inclusive) or the path of a size-predictor checkpoint
    (generate.py:69-99).  A ``SizeClassifier`` instance is accepted in place of the path."""
    if isinstance(linker_size, SizeClassifier):
        size_nn = linker_size.eval().to(device)
        return lambda _data: size_nn.sample_sizes(_data, with_pocket=with_pocket)
    linker_size = str(linker_size)
    if linker_size.isdigit():
        size = int(linker_size)
        return lambda _data: torch.ones(_data['positions'].shape[0], device=device, dtype=const.TORCH_INT) * size
    boundaries = [x.strip() for x in linker_size.split(',')]
    if len(boundaries) == 2 and boundaries[0].isdigit() and boundaries[1].isdigit():
        left, right = int(boundaries[0]), int(boundaries[1])
        return lambda _data: torch.randint(left, right + 1, (len(_data['positions']),), device=device,
                                           dtype=const.TORCH_INT)
    size_nn = SizeClassifier.load_from_checkpoint(linker_size, map_location=device).eval().to(device)
    return lambda _data: size_nn.sample_sizes(_data, with_pocket=with_pocket)


def _load_ddpm(model, device, n_steps):
    ddpm = model if isinstance(model, DDPM) else DDPM.load_from_checkpoint(model, map_location=device)
    ddpm = ddpm.eval().to(device)
    if n_steps is not None:
        ddpm.edm.T = n_steps
    return ddpm


def _anchor_flags(charges, anchors):
    flags = np.zeros_like(charges)
    if anchors is not None:
        for anchor in str(anchors).split(','):
            flags[int(anchor.strip()) - 1] = 1
    return flags


def _batches(dataset, batch_size, collate_fn):
    for start in range(0, len(dataset), batch_size):
        yield collate_fn(dataset[start:start + batch_size])


def _sample_and_save(ddpm, dataset, collate_fn, sample_fn, batch_size, output_dir, name, com_key, hide_pocket):
    written = []
    for batch_i, data in enumerate(_batches(dataset, batch_size, collate_fn)):
        n = len(data['positions'])
        chain = None
        for _ in range(5):                                    # generate.py:152-160
            try:
                chain, node_mask = ddpm.sample_chain(data, sample_fn=sample_fn, keep_frames=1)
                break
            except FoundNaNException:
                continue
        if chain is None:
            raise Exception('Could not generate in 5 attempts')
        x = chain[0][:, :, :ddpm.n_dims]
        h = chain[0][:, :, ddpm.n_dims:]
        # put the molecule back to the initial frame (generate.py:165-170): the chain lives in the COM frame of the
        # fragments / anchors; `mean` broadcasts over the template width (linker rows included)
        com_mask = data[com_key] if ddpm.center_of_mass == 'fragments' else data['anchors']
        pos_masked = data['positions'] * com_mask
        mean = torch.sum(pos_masked, dim=1, keepdim=True) / com_mask.sum(1, keepdims=True)
        x = x + mean * node_mask
        offset = batch_i * batch_size
        names = [f'output_{offset + i}_{name}' for i in range(n)]
        if hide_pocket:                                       # generate_with_pocket.py:272
            node_mask = node_mask.clone()
            width = data['pocket_mask'].shape[1]
            node_mask[:, :width][data['pocket_mask'].bool()] = 0
        save_xyz_file(output_dir, h, x, node_mask, names=names, is_geom=ddpm.is_geom, suffix='')
        written += [os.path.join(output_dir, f'{nm}_.xyz') for nm in names]
    return written


def generate(input_path, model, output_dir, n_samples, n_steps, linker_size, anchors=None, device=None):
    """``generate.py`` main(): fragments file -> ``n_samples`` molecules with a sampled linker, as ``.xyz`` files."""
    device = torch.device(device or ('cuda' if torch.cuda.is_available() else 'cpu'))
    os.makedirs(output_dir, exist_ok=True)
    sample_fn = make_sample_fn(linker_size, device)
    ddpm = _load_ddpm(model, device, n_steps)
    if ddpm.center_of_mass == 'anchors' and anchors is None:
        raise ValueError('Please pass anchor atoms indices or use another DiffLinker model that does not require '
                         'information about anchors')
    if input_path.split('.')[-1] not in ['sdf', 'pdb', 'mol', 'mol2']:
        raise ValueError('Please upload the file in one of the following formats: .pdb, .sdf, .mol, .mol2')
    molecule = read_molecule(input_path)
    name = '.'.join(input_path.split('/')[-1].split('.')[:-1])
    positions, one_hot, charges = parse_molecule(molecule, is_geom=ddpm.is_geom)
    t = lambda a: torch.tensor(a, dtype=const.TORCH_FLOAT, device=device)     # noqa: E731
    dataset = [{
        'uuid': '0', 'name': '0', 'positions': t(positions), 'one_hot': t(one_hot), 'charges': t(charges),
        'anchors': t(_anchor_flags(charges, anchors)), 'fragment_mask': t(np.ones_like(charges)),
        'linker_mask': t(np.zeros_like(charges)), 'num_atoms': len(positions),
    }] * n_samples
    return _sample_and_save(ddpm, dataset, collate_with_fragment_edges, sample_fn, min(n_samples, 64), output_dir, name,
                            com_key='fragment_mask', hide_pocket=False)


def _generate_pocket_common(frag, pocket, ddpm, sample_fn, output_dir, name, n_samples, anchors, max_batch_size, device):
    frag_pos, frag_one_hot, frag_charges = frag
    pocket_pos, pocket_one_hot, pocket_charges = pocket
    positions = np.concatenate([frag_pos, pocket_pos], axis=0)
    one_hot = np.concatenate([frag_one_hot, pocket_one_hot], axis=0)
    charges = np.concatenate([frag_charges, pocket_charges], axis=0)
    ones_f, zeros_f = np.ones_like(frag_charges), np.zeros_like(frag_charges)
    ones_p, zeros_p = np.ones_like(pocket_charges), np.zeros_like(pocket_charges)
    t = lambda a: torch.tensor(a, dtype=const.TORCH_FLOAT, device=device)     # noqa: E731
    dataset = [{
        'uuid': '0', 'name': '0', 'positions': t(positions), 'one_hot': t(one_hot), 'charges': t(charges),
        'anchors': t(_anchor_flags(charges, anchors)),
        'fragment_only_mask': t(np.concatenate([ones_f, zeros_p])), 'pocket_mask': t(np.concatenate([zeros_f, ones_p])),
        'fragment_mask': t(np.concatenate([ones_f, ones_p])), 'linker_mask': t(np.concatenate([zeros_f, zeros_p])),
        'num_atoms': len(positions),
    }] * n_samples
    dataset = MOADDataset(data=dataset)                   # generate_with_pocket.py:249-250: DDPM.sample_chain keys the
    ddpm.val_dataset = dataset                            # centre-of-mass mask on the dataset type (lightning.py:443)
    return _sample_and_save(ddpm, dataset, collate_with_fragment_without_pocket_edges, sample_fn,
                            min(n_samples, max_batch_size), output_dir, name, com_key='fragment_only_mask',
                            hide_pocket=True)


def generate_with_pocket(input_path, pocket_path, backbone_atoms_only, model, output_dir, n_samples, n_steps, linker_size,
                         anchors=None, max_batch_size=64, random_seed=None, device=None):
    """``generate_with_pocket.py`` main(): the pocket is given as its own PDB file."""
    device = torch.device(device or ('cuda' if torch.cuda.is_available() else 'cpu'))
    os.makedirs(output_dir, exist_ok=True)
    if random_seed is not None:
        set_deterministic(random_seed)
    sample_fn = make_sample_fn(linker_size, device, with_pocket=True)
    ddpm = _load_ddpm(model, device, n_steps)
    if ddpm.center_of_mass == 'anchors' and anchors is None:
        raise ValueError('Please pass anchor atoms indices or use another DiffLinker model that does not require '
                         'information about anchors')
    if input_path.split('.')[-1] not in ['sdf', 'pdb', 'mol', 'mol2']:
        raise ValueError('Please upload the fragments file in one of the following formats: .pdb, .sdf, .mol, .mol2')
    if pocket_path.split('.')[-1] != 'pdb':
        raise ValueError('Please upload the pocket file in .pdb format')
    molecule = read_molecule(input_path)
    name = '.'.join(input_path.split('/')[-1].split('.')[:-1])
    frag = parse_molecule(molecule, is_geom=ddpm.is_geom)
    pocket = pocket_arrays(read_pocket(pocket_path), backbone_atoms_only)
    return _generate_pocket_common(frag, pocket, ddpm, sample_fn, output_dir, name, n_samples, anchors, max_batch_size,
                                   device)


def generate_with_protein(input_path, protein_path, backbone_atoms_only, model, output_dir, n_samples, n_steps,
                          linker_size, anchors=None, max_batch_size=64, random_seed=None, device=None):
    """``generate_with_protein.py`` main(): the pocket = residues of the protein within 6 A of the fragments."""
    device = torch.device(device or ('cuda' if torch.cuda.is_available() else 'cpu'))
    os.makedirs(output_dir, exist_ok=True)
    if random_seed is not None:
        set_deterministic(random_seed)
    sample_fn = make_sample_fn(linker_size, device, with_pocket=True)
    ddpm = _load_ddpm(model, device, n_steps)
    if ddpm.center_of_mass == 'anchors' and anchors is None:
        raise ValueError('Please pass anchor atoms indices or use another DiffLinker model that does not require '
                         'information about anchors')
    molecule = read_molecule(input_path)
    name = '.'.join(input_path.split('/')[-1].split('.')[:-1])
    frag = parse_molecule(molecule, is_geom=ddpm.is_geom)
    pocket = get_pocket(molecule, protein_path, backbone_atoms_only)
    return _generate_pocket_common(frag, pocket, ddpm, sample_fn, output_dir, name, n_samples, anchors, max_batch_size,
                                   device)


def main(argv=None):
    p = argparse.ArgumentParser(description='DiffLinker sampling on MI355X (generate.py / generate_with_pocket.py / '
                                            'generate_with_protein.py of the reference)')
    p.add_argument('--fragments', required=True, help='file with the input fragments (.sdf .mol .mol2 .pdb)')
    p.add_argument('--pocket', default=None, help='PDB file of the pocket residues (generate_with_pocket.py)')
    p.add_argument('--protein', default=None, help='PDB file of the whole protein (generate_with_protein.py)')
    p.add_argument('--backbone_atoms_only', action='store_true', default=False)
    p.add_argument('--model', required=True, help='DiffLinker checkpoint')
    p.add_argument('--linker_size', required=True, help='integer, "lo,hi", or a size-predictor checkpoint')
    p.add_argument('--output', default='./')
    p.add_argument('--n_samples', type=int, default=5)
    p.add_argument('--n_steps', type=int, default=None)
    p.add_argument('--anchors', default=None, help='comma-separated 1-based indices of the anchor atoms')
    p.add_argument('--max_batch_size', type=int, default=64)
    p.add_argument('--random_seed', type=int, default=None)
    a = p.parse_args(argv)
    if a.pocket is not None:
        files = generate_with_pocket(a.fragments, a.pocket, a.backbone_atoms_only, a.model, a.output, a.n_samples,
                                     a.n_steps, a.linker_size, a.anchors, a.max_batch_size, a.random_seed)
    elif a.protein is not None:
        files = generate_with_protein(a.fragments, a.protein, a.backbone_atoms_only, a.model, a.output, a.n_samples,
                                      a.n_steps, a.linker_size, a.anchors, a.max_batch_size, a.random_seed)
    else:
        files = generate(a.fragments, a.model, a.output, a.n_samples, a.n_steps, a.linker_size, a.anchors)
    print(f'Saved {len(files)} generated molecules in .xyz format in directory {a.output}')


if __name__ == '__main__':
    main()
