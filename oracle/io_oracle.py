"""Oracle for the I/O shell of the generation scripts (row f2): an independent restatement of what the reference's
helpers compute from the files — WITHOUT RDKit / Bio.PDB, which this image lacks, so parity with those libraries
themselves stays UNPINNED (stated in DESIGN.md); what is pinned is the arithmetic of the scripts on top of them:

* ``heavy_atoms_of_sdf``  — ``read_molecule`` + ``parse_molecule`` of generate.py:50-59 / generate_with_protein.py:63-82:
  first record of an SDF (V2000 counts line, fixed columns), hydrogens removed, symbols and coordinates in file order;
* ``pocket_of_protein``   — ``get_pocket`` of generate_with_protein.py:85-148: residues (matched by residue NUMBER
  alone, :97,111) with an atom within 6 A of a fragment atom, all their atoms (or N/CA/C/O) of the model's vocabulary.

``src/visualizer.py::save_xyz_file`` needs neither library: it is run itself by tests/golden/make_golden.py.
TEST INFRASTRUCTURE — see ``oracle/__init__.py``.
"""
import numpy as np

GEOM_ATOMS = {'C': 6, 'O': 8, 'N': 7, 'F': 9, 'S': 16, 'Cl': 17, 'Br': 35, 'I': 53, 'P': 15}     # const.py:29-37


def heavy_atoms_of_sdf(path):
    rows = open(path).read().split('\n')
    n_atoms = int(rows[3][0:3])
    symbols, coords = [], []
    for k in range(n_atoms):
        row = rows[4 + k]
        sym = row[31:34].strip()
        if sym == 'H':
            continue
        symbols.append(sym)
        coords.append((float(row[0:10]), float(row[10:20]), float(row[20:30])))
    return symbols, np.array(coords)


def pocket_of_protein(pdb_path, fragment_coords, backbone_atoms_only=False):
    names, elements, resnums, xyz = [], [], [], []
    for row in open(pdb_path).read().split('\n'):
        if row[0:6] not in ('ATOM  ', 'HETATM'):
            continue
        if row[16] not in (' ', 'A'):                       # fixtures carry no second alternate location worth keeping
            continue
        names.append(row[12:16].strip())
        el = row[76:78].strip() or row[12:14].strip()
        elements.append(el[0].upper() + el[1:].lower())
        resnums.append(int(row[22:26]))
        xyz.append((float(row[30:38]), float(row[38:46]), float(row[46:54])))
    xyz32 = np.array(xyz, dtype=np.float32)                 # Bio.PDB keeps float32 coordinates
    resnums = np.array(resnums)
    close = np.zeros(len(xyz32), dtype=bool)
    for f in np.asarray(fragment_coords, dtype=np.float64):
        close |= np.sqrt(((xyz32.astype(np.float64) - f) ** 2).sum(1)) <= 6
    contact = set(resnums[close].tolist())
    pos, sym = [], []
    for k in range(len(xyz32)):
        if resnums[k] not in contact:
            continue
        if backbone_atoms_only and names[k] not in ('N', 'CA', 'C', 'O'):
            continue
        if elements[k] not in GEOM_ATOMS:
            continue
        pos.append(xyz32[k])
        sym.append(elements[k])
    return np.array(pos), sym
