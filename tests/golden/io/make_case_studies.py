"""Copy the reference's real case-study inputs into small fixtures (build container only; reads /root/reference):
the fragment SDF files as they are, the protein PDB files trimmed to the residues that have an atom within 12 A of a
fragment atom (the 6 A pocket selection of generate_with_protein.py:85-148 then still has residues to reject).
    python tests/golden/io/make_case_studies.py"""
import os
import shutil

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference/case_studies'
CASES = {'hsp90': ('3hz1_modified_fragments_obabel.sdf', '3hz1_protein.pdb'),
         'jnk': ('3fi3_fragments.sdf', '3fi3_protein.pdb')}


def sdf_coords(path):
    lines = open(path).read().splitlines()
    n = int(lines[3][:3])
    return np.array([[float(lines[4 + k][10 * d:10 * d + 10]) for d in range(3)] for k in range(n)])


for case, (sdf, pdb) in CASES.items():
    out = os.path.join(HERE, 'case_studies')
    os.makedirs(out, exist_ok=True)
    shutil.copy(os.path.join(REF, case, sdf), os.path.join(out, f'{case}_fragments.sdf'))
    os.chmod(os.path.join(out, f'{case}_fragments.sdf'), 0o644)
    frag = sdf_coords(os.path.join(REF, case, sdf))
    atoms = [ln for ln in open(os.path.join(REF, case, pdb)).read().splitlines() if ln.startswith(('ATOM', 'HETATM'))]
    xyz = np.array([[float(ln[30:38]), float(ln[38:46]), float(ln[46:54])] for ln in atoms])
    key = [(ln[21], ln[22:27]) for ln in atoms]                       # chain, residue number + insertion code
    near = np.linalg.norm(xyz[:, None, :] - frag[None, :, :], axis=-1).min(1) <= 12.0
    keep = {k for k, n_ in zip(key, near) if n_}
    kept = [ln for ln, k in zip(atoms, key) if k in keep]
    with open(os.path.join(out, f'{case}_protein_12A.pdb'), 'w') as f:
        f.write('\n'.join(kept) + '\nEND\n')
    print(case, len(frag), 'fragment atoms;', len(kept), 'of', len(atoms), 'protein atoms kept')
