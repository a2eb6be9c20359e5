"""Molecule I/O of the generation scripts without RDKit / Biopython / OpenBabel.

The reference reads its inputs with ``rdkit.Chem`` (``generate.py:50-59``) and ``Bio.PDB.PDBParser``
(``generate_with_pocket.py:84-113``, ``generate_with_protein.py:85-148``) and writes ``.xyz`` files with
``src/visualizer.py:14-31``.  None of those packages exist in this image, and the sampling path only needs element
symbols and coordinates, so the few record types involved are parsed here directly:

* ``read_molecule``: first molecule of an ``.sdf`` / ``.mol`` (V2000 and V3000 atom blocks), ``.pdb`` (first model,
  first alternate location), ``.mol2`` (``@<TRIPOS>ATOM``) or ``.xyz`` file, hydrogens removed
  (``removeHs=True`` + ``Chem.RemoveAllHs``, generate.py:53-59,117);
* ``parse_molecule``: positions / one-hot / charges with the reference's vocabularies (src/datasets.py:22-37);
* ``read_pocket`` / ``get_pocket``: the pocket dictionaries of the two pocket scripts, including their quirks
  (every model is walked, the highest-occupancy alternate location is kept, contact residues are matched by residue
  NUMBER only, the 'full' atom list keeps whatever elements the file has);
* ``save_xyz_file`` / ``load_xyz_files`` / ``load_molecule_xyz``: byte-compatible with src/visualizer.py:14-59.

Bond perception (``obabel xyz -> sdf``, generate.py:163-166) is post-processing outside the sampling path and is
not provided.
"""
import os
from dataclasses import dataclass, field
from typing import List

import numpy as np
import torch

from . import const

_TWO_LETTER = {'CL': 'Cl', 'BR': 'Br', 'NA': 'Na', 'MG': 'Mg', 'ZN': 'Zn', 'FE': 'Fe', 'CA': 'Ca', 'MN': 'Mn',
               'CU': 'Cu', 'SE': 'Se', 'SI': 'Si', 'LI': 'Li', 'CO': 'Co', 'NI': 'Ni', 'AL': 'Al'}


def _symbol(raw):
    s = raw.strip()
    if not s:
        raise ValueError('empty element symbol')
    up = s.upper()
    if up in _TWO_LETTER:
        return _TWO_LETTER[up]
    return up[0] + up[1:].lower()


@dataclass
class Molecule:
    """Heavy atoms of one molecule: what ``parse_molecule`` needs from an RDKit ``Mol``."""
    symbols: List[str]
    positions: np.ndarray                      # [n, 3] float64, file order
    name: str = ''
    props: dict = field(default_factory=dict)

    def __len__(self):
        return len(self.symbols)


def _without_hydrogens(symbols, coords, name):
    keep = [i for i, s in enumerate(symbols) if s not in ('H', 'D', 'T')]
    pos = np.asarray([coords[i] for i in keep], dtype=np.float64).reshape(len(keep), 3)
    return Molecule([symbols[i] for i in keep], pos, name)


def _read_molblock(lines):
    """Atom block of the first molecule of an SDF / MOL file (CTfile V2000 or V3000)."""
    if len(lines) < 4:
        raise ValueError('truncated mol block')
    name = lines[0].strip()
    counts = lines[3]
    symbols, coords = [], []
    if 'V3000' in counts:
        in_atoms = False
        for ln in lines[4:]:
            t = ln.strip()
            if t.startswith('M  V30 BEGIN ATOM'):
                in_atoms = True
            elif t.startswith('M  V30 END ATOM'):
                break
            elif in_atoms and t.startswith('M  V30'):
                parts = t.split()
                symbols.append(_symbol(parts[3]))
                coords.append([float(parts[4]), float(parts[5]), float(parts[6])])
    else:
        try:
            n_atoms = int(counts[0:3])
        except ValueError as e:
            raise ValueError(f'bad counts line: {counts!r}') from e
        for ln in lines[4:4 + n_atoms]:
            coords.append([float(ln[0:10]), float(ln[10:20]), float(ln[20:30])])
            symbols.append(_symbol(ln[31:34]))
        if len(symbols) != n_atoms:
            raise ValueError('truncated atom block')
    return symbols, coords, name


def _pdb_element(line):
    el = line[76:78].strip() if len(line) >= 78 else ''
    if el and el.isalpha():
        return _symbol(el)
    name = line[12:16]
    # no element column: columns 13-14 right-justify the symbol (" CA " is carbon alpha, "CA  " is calcium)
    guess = name[:2].strip() if not name[0].isdigit() and name[0] != ' ' and name[:2].upper() in _TWO_LETTER else \
        ''.join(c for c in name if c.isalpha())[:1]
    return _symbol(guess)


def _read_pdb_ligand(lines):
    symbols, coords = [], []
    for ln in lines:
        rec = ln[0:6]
        if rec.startswith('ENDMDL'):
            break
        if rec in ('ATOM  ', 'HETATM'):
            if ln[16] not in (' ', 'A'):
                continue
            symbols.append(_pdb_element(ln))
            coords.append([float(ln[30:38]), float(ln[38:46]), float(ln[46:54])])
    return symbols, coords


def _read_mol2(lines):
    symbols, coords, name = [], [], ''
    section = None
    for k, ln in enumerate(lines):
        t = ln.strip()
        if t.startswith('@<TRIPOS>'):
            if section == 'ATOM':
                break
            section = t[len('@<TRIPOS>'):]
            if section == 'MOLECULE' and k + 1 < len(lines):
                name = lines[k + 1].strip()
            continue
        if section == 'ATOM' and t:
            parts = t.split()
            coords.append([float(parts[2]), float(parts[3]), float(parts[4])])
            symbols.append(_symbol(parts[5].split('.')[0]))
    return symbols, coords, name


def read_molecule(path):
    """First molecule of ``path`` with the hydrogens removed (generate.py:50-59 + :117)."""
    if path.split('.')[-1] not in ('pdb', 'mol', 'sdf', 'mol2', 'xyz'):
        raise Exception('Unknown file extension')
    with open(path) as f:
        lines = f.read().splitlines()
    base = '.'.join(os.path.basename(path).split('.')[:-1])
    if path.endswith('.pdb'):
        symbols, coords = _read_pdb_ligand(lines)
        name = base
    elif path.endswith('.mol') or path.endswith('.sdf'):
        symbols, coords, name = _read_molblock(lines)
    elif path.endswith('.mol2'):
        symbols, coords, name = _read_mol2(lines)
    elif path.endswith('.xyz'):
        n = int(lines[0].split()[0])
        symbols = [_symbol(ln.split()[0]) for ln in lines[2:2 + n]]
        coords = [[float(v) for v in ln.split()[1:4]] for ln in lines[2:2 + n]]
        name = base
    else:
        raise Exception('Unknown file extension')
    if not symbols:
        raise ValueError(f'no atoms found in {path}')
    return _without_hydrogens(symbols, coords, name or base)


def get_one_hot(atom, atoms_dict):
    one_hot = np.zeros(len(atoms_dict))
    one_hot[atoms_dict[atom]] = 1
    return one_hot


def parse_molecule(mol, is_geom):
    """``(positions [n,3], one_hot [n,types], charges [n])`` — src/datasets.py:28-37."""
    atom2idx = const.GEOM_ATOM2IDX if is_geom else const.ATOM2IDX
    charges_dict = const.GEOM_CHARGES if is_geom else const.CHARGES
    one_hot, charges = [], []
    for s in mol.symbols:
        if s not in atom2idx:
            raise KeyError(f'element {s!r} is outside the model vocabulary {sorted(atom2idx)}')
        one_hot.append(get_one_hot(s, atom2idx))
        charges.append(charges_dict[s])
    return np.asarray(mol.positions, dtype=np.float64), np.array(one_hot), np.array(charges)


# ---------------------------------------------------------------------------------------------------
# protein pockets
@dataclass
class _PdbAtom:
    name: str
    element: str
    coord: List[float]
    resseq: int
    occupancy: float


def _walk_pdb(path):
    """Atoms in Bio.PDB iteration order (models, chains, residues, atoms as they appear); for an atom with alternate
    locations one entry at the position of its first record, holding the location with the highest occupancy."""
    atoms, index = [], {}
    model = 0
    with open(path) as f:
        for ln in f:
            rec = ln[0:6]
            if rec.startswith('ENDMDL'):
                model += 1
                continue
            if rec not in ('ATOM  ', 'HETATM'):
                continue
            name = ln[12:16].strip()
            try:
                occ = float(ln[54:60])
            except ValueError:
                occ = 1.0
            atom = _PdbAtom(name, _pdb_element(ln).upper(), [float(ln[30:38]), float(ln[38:46]), float(ln[46:54])],
                            int(ln[22:26]), occ)
            key = (model, ln[21], ln[22:27], ln[17:20], name)      # chain, resseq + icode, resname, atom
            if ln[16] != ' ' and key in index:
                if occ > atoms[index[key]].occupancy:
                    atoms[index[key]] = atom
                continue
            if ln[16] != ' ':
                index[key] = len(atoms)
            atoms.append(atom)
    return atoms


def read_pocket(path):
    """Pocket dictionary of generate_with_pocket.py:84-113 (the file already holds the pocket residues only)."""
    full_c, full_t, bb_c, bb_t = [], [], [], []
    for a in _walk_pdb(path):
        full_c.append(a.coord)
        full_t.append(a.element)
        if a.name == 'H':
            continue
        if a.name in {'N', 'CA', 'C', 'O'}:
            bb_c.append(a.coord)
            bb_t.append(a.element)
    return {'full_coord': np.array(full_c), 'full_types': np.array(full_t),
            'bb_coord': np.array(bb_c), 'bb_types': np.array(bb_t)}


def get_pocket(mol, pdb_path, backbone_atoms_only=False):
    """Residues with an atom within 6 A of the ligand (generate_with_protein.py:85-148).  Contact residues are matched
    by residue NUMBER alone, as the reference does: equal numbers in other chains are included."""
    atoms = _walk_pdb(pdb_path)
    residue_ids = np.array([a.resseq for a in atoms])
    atom_coords = np.array([a.coord for a in atoms], dtype=np.float32)       # Bio.PDB stores float32 coordinates
    mol_coords = np.asarray(mol.positions if isinstance(mol, Molecule) else mol, dtype=np.float64)
    distances = np.linalg.norm(atom_coords[:, None, :] - mol_coords[None, :, :], axis=-1)
    contact = set(np.unique(residue_ids[np.where(distances.min(1) <= 6)[0]]).tolist())
    pos, one_hot, charges = [], [], []
    for a in atoms:
        if a.resseq not in contact:
            continue
        if backbone_atoms_only and a.name not in {'N', 'CA', 'C', 'O'}:
            continue
        sym = _symbol(a.element)
        if a.element not in const.GEOM_ATOM2IDX and sym not in const.GEOM_ATOM2IDX:
            continue
        key = a.element if a.element in const.GEOM_ATOM2IDX else sym
        pos.append(a.coord)
        one_hot.append(get_one_hot(key, const.GEOM_ATOM2IDX))
        charges.append(const.GEOM_CHARGES[key])
    return np.array(pos), np.array(one_hot), np.array(charges)


def pocket_arrays(pocket_data, backbone_atoms_only):
    """One-hot / charges of a ``read_pocket`` dictionary (generate_with_pocket.py:200-209); unknown elements raise like
    the reference's dictionary lookup does."""
    mode = 'bb' if backbone_atoms_only else 'full'
    one_hot, charges = [], []
    for t in pocket_data[f'{mode}_types']:
        key = t if t in const.GEOM_ATOM2IDX else _symbol(t)
        if key not in const.GEOM_ATOM2IDX:
            raise KeyError(f'pocket element {t!r} is outside the model vocabulary {sorted(const.GEOM_ATOM2IDX)}')
        one_hot.append(get_one_hot(key, const.GEOM_ATOM2IDX))
        charges.append(const.GEOM_CHARGES[key])
    return pocket_data[f'{mode}_coord'], np.array(one_hot), np.array(charges)


# ---------------------------------------------------------------------------------------------------
# xyz output
def save_xyz_file(path, one_hot, positions, node_mask, names, is_geom, suffix=''):
    """One ``<name>_<suffix>.xyz`` per molecule, masked atoms only, ``%.9f`` coordinates (src/visualizer.py:14-31)."""
    idx2atom = const.GEOM_IDX2ATOM if is_geom else const.IDX2ATOM
    one_hot, positions, node_mask = one_hot.detach().cpu(), positions.detach().cpu(), node_mask.detach().cpu()
    for batch_i in range(one_hot.size(0)):
        mask = node_mask[batch_i].reshape(-1)
        atom_idx = torch.where(mask)[0]
        atoms = torch.argmax(one_hot[batch_i], dim=1)
        with open(os.path.join(path, f'{names[batch_i]}_{suffix}.xyz'), 'w') as f:
            f.write('%d\n\n' % int(mask.sum()))
            for atom_i in atom_idx:
                f.write('%s %.9f %.9f %.9f\n' % (idx2atom[atoms[atom_i].item()], positions[batch_i, atom_i, 0],
                                                 positions[batch_i, atom_i, 1], positions[batch_i, atom_i, 2]))


def load_xyz_files(path, suffix=''):
    """Paths of the ``*_<suffix>.xyz`` files of a directory, latest frame index first (src/visualizer.py:34-40)."""
    files = [fname for fname in os.listdir(path) if fname.endswith(f'_{suffix}.xyz')]
    files = sorted(files, key=lambda f: -int(f.replace(f'_{suffix}.xyz', '').split('_')[-1]))
    return [os.path.join(path, fname) for fname in files]


def load_molecule_xyz(file, is_geom):
    """``(positions [n,3], one_hot [n,types], charges [n,1] zeros)`` of an ``.xyz`` written by ``save_xyz_file``
    (src/visualizer.py:43-59)."""
    atom2idx = const.GEOM_ATOM2IDX if is_geom else const.ATOM2IDX
    with open(file, encoding='utf8') as f:
        n_atoms = int(f.readline())
        one_hot = torch.zeros(n_atoms, len(atom2idx))
        charges = torch.zeros(n_atoms, 1)
        positions = torch.zeros(n_atoms, 3)
        f.readline()
        atoms = f.readlines()
        for i in range(n_atoms):
            parts = atoms[i].split(' ')
            one_hot[i, atom2idx[parts[0]]] = 1
            positions[i, :] = torch.Tensor([float(e) for e in parts[1:]])
    return positions, one_hot, charges
