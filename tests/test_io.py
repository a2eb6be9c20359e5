"""CPU tests of the RDKit-free molecule I/O (difflinker_amd/io.py) on hand-made fixtures (tests/golden/io/) and,
in the build container only, on the reference's case-study files."""
import os

import numpy as np
import pytest
import torch

from difflinker_amd import const, io

HERE = os.path.dirname(os.path.abspath(__file__))
IO_DIR = os.path.join(HERE, 'golden', 'io')
CASES = '/root/reference/case_studies'


def test_sdf_first_record_without_hydrogens():
    m = io.read_molecule(os.path.join(IO_DIR, 'frag.sdf'))
    assert m.symbols == ['C', 'C', 'N', 'O', 'C', 'Cl', 'S'] and m.name == 'two_fragments'
    assert np.allclose(m.positions[0], [1.2, 0.0, 0.1]) and np.allclose(m.positions[5], [-5.2, 1.1, 0.6])
    pos, one_hot, charges = io.parse_molecule(m, is_geom=False)
    assert pos.shape == (7, 3) and one_hot.shape == (7, 8)
    assert one_hot.argmax(1).tolist() == [0, 0, 2, 1, 0, 5, 4] and charges.tolist() == [6, 6, 7, 8, 6, 17, 16]
    assert io.parse_molecule(m, is_geom=True)[1].shape == (7, 9)


def test_v3000_mol2_pdb_and_vocabulary_errors():
    m = io.read_molecule(os.path.join(IO_DIR, 'frag_v3000.mol'))
    assert m.symbols == ['C', 'Br', 'P'] and np.allclose(m.positions[2], [-4.0, 0.5, 0.0])
    with pytest.raises(KeyError, match="'P'"):
        io.parse_molecule(m, is_geom=False)                  # phosphorus only exists in the GEOM vocabulary
    assert io.parse_molecule(m, is_geom=True)[2].tolist() == [6, 35, 15]
    m2 = io.read_molecule(os.path.join(IO_DIR, 'lig.mol2'))
    assert m2.symbols == ['C', 'N', 'Cl'] and m2.name == 'mol2_example'
    mp = io.read_molecule(os.path.join(IO_DIR, 'lig.pdb'))
    assert mp.symbols == ['C', 'Cl', 'N']                    # element column, atom-name fallback, altloc A only
    assert np.allclose(mp.positions[2], [2.5, 0.3, -0.2])
    with pytest.raises(Exception, match='Unknown file extension'):
        io.read_molecule(os.path.join(IO_DIR, 'frag.smi'))


def test_pocket_extraction_rules():
    frag = io.read_molecule(os.path.join(IO_DIR, 'frag.sdf'))
    prot = os.path.join(IO_DIR, 'protein.pdb')
    pos, one_hot, charges = io.get_pocket(frag, prot)
    # residue 10 of chain A is in contact; chain B's residue 10 rides along (matched by number only, like the
    # reference); the zinc ion is in contact but outside the vocabulary; the water oxygen (residue 301) is kept
    assert pos.shape == (8, 3)
    assert charges.tolist() == [7, 6, 6, 8, 7, 6, 16, 8]
    bb = io.get_pocket(frag, prot, backbone_atoms_only=True)
    assert bb[2].tolist() == [7, 6, 6, 8, 7, 6, 8]           # N CA C O + N CA of chain B + the water 'O'
    d = io.read_pocket(prot)
    assert d['full_coord'].shape == (13, 3)                  # 14 records, the two CB alternates are one atom
    assert d['full_types'].tolist().count('ZN') == 1
    cb = d['full_coord'][6]
    assert np.allclose(cb, [32.5, 31.5, 31.5])               # the higher-occupancy location B
    assert d['bb_types'].tolist() == ['N', 'C', 'C', 'O', 'N', 'C', 'N', 'C', 'O']
    with pytest.raises(KeyError, match='ZN'):
        io.pocket_arrays(d, backbone_atoms_only=False)
    p, oh, ch = io.pocket_arrays(d, backbone_atoms_only=True)
    assert p.shape == (9, 3) and oh.shape == (9, 9)


def test_xyz_roundtrip_and_format(tmp_path):
    one_hot = torch.eye(9)[[0, 5, 2, 8]].unsqueeze(0)
    positions = torch.tensor([[[0.1, -2.5, 3.0], [1.0, 2.0, 3.0], [9.0, 9.0, 9.0], [-1.25, 0.0, 4.5]]])
    node_mask = torch.tensor([[[1], [1], [0], [1]]], dtype=const.TORCH_INT)
    io.save_xyz_file(str(tmp_path), one_hot, positions, node_mask, names=['mol_7'], is_geom=True, suffix='')
    path = os.path.join(tmp_path, 'mol_7_.xyz')
    text = open(path).read()
    assert text == '3\n\nC 0.100000001 -2.500000000 3.000000000\nCl 1.000000000 2.000000000 3.000000000\n' \
                   'P -1.250000000 0.000000000 4.500000000\n'
    pos, oh, ch = io.load_molecule_xyz(path, is_geom=True)
    assert torch.allclose(pos, positions[0][[0, 1, 3]]) and oh.argmax(1).tolist() == [0, 5, 8] and ch.shape == (3, 1)
    assert io.load_xyz_files(str(tmp_path)) == [path]
    back = io.read_molecule(path)
    assert back.symbols == ['C', 'Cl', 'P']


@pytest.mark.skipif(not os.path.isdir(CASES), reason='reference tree only exists in the build container')
def test_reference_case_study_files_parse():
    frag = io.read_molecule(os.path.join(CASES, 'hsp90', '3hz1_original_fragments.sdf'))
    assert len(frag) == 27 and set(frag.symbols) <= set(const.GEOM_ATOM2IDX)
    pos, one_hot, charges = io.get_pocket(frag, os.path.join(CASES, 'hsp90', '3hz1_protein.pdb'))
    assert pos.shape[0] == one_hot.shape[0] == charges.shape[0] and 100 < pos.shape[0] < 400
    # every kept atom belongs to a residue with an atom within 6 A: the closest pocket atom is closer than that
    d = np.linalg.norm(pos[:, None] - frag.positions[None], axis=-1)
    assert d.min() <= 6.0
    bb = io.get_pocket(frag, os.path.join(CASES, 'hsp90', '3hz1_protein.pdb'), backbone_atoms_only=True)
    assert bb[0].shape[0] < pos.shape[0]
    for rel in ('impdh/5ou2_fragments_input.sdf', 'jnk/3fi3_fragments.sdf', 'jnk/3fi3_linker.sdf'):
        m = io.read_molecule(os.path.join(CASES, rel))
        io.parse_molecule(m, is_geom=True)


def test_generation_driver_helpers(tmp_path):
    """Host-side pieces of the generation drivers (generate.py:69-99, :128-146): the three kinds of ``--linker_size``,
    1-based anchor indices, batching, the pocket collate's fragment-only edge mask."""
    from difflinker_amd.datasets import collate_with_fragment_without_pocket_edges
    from difflinker_amd.generate import _anchor_flags, _batches, make_sample_fn
    data = {'positions': torch.zeros(5, 7, 3)}
    fixed = make_sample_fn('6', 'cpu')(data)
    assert fixed.dtype == const.TORCH_INT and fixed.tolist() == [6] * 5
    torch.manual_seed(0)
    uni = make_sample_fn('3, 9', 'cpu')(data)
    assert uni.shape == (5,) and int(uni.min()) >= 3 and int(uni.max()) <= 9
    with pytest.raises(FileNotFoundError):
        make_sample_fn(os.path.join(tmp_path, 'missing.ckpt'), 'cpu')
    assert _anchor_flags(np.zeros(6), '2, 5').tolist() == [0, 1, 0, 0, 1, 0]
    assert _anchor_flags(np.zeros(3), None).tolist() == [0, 0, 0]
    assert [len(b) for b in _batches(list(range(10)), 4, lambda x: x)] == [4, 4, 2]
    item = {'positions': torch.randn(5, 3), 'one_hot': torch.eye(9)[:5], 'anchors': torch.zeros(5),
            'fragment_only_mask': torch.tensor([1., 1, 0, 0, 0]), 'pocket_mask': torch.tensor([0., 0, 1, 1, 0]),
            'fragment_mask': torch.tensor([1., 1, 1, 1, 0]), 'linker_mask': torch.tensor([0., 0, 0, 0, 1]),
            'num_atoms': 5, 'uuid': 0, 'name': 'm'}
    out = collate_with_fragment_without_pocket_edges([item])
    em = out['edge_mask'].view(5, 5)
    assert em[:2, :2].tolist() == [[-2, -1], [-1, -2]] and float(em[2:].abs().sum() + em[:, 2:].abs().sum()) == 0.0
    assert out['atom_mask'].view(-1).tolist() == [1, 1, 1, 1, 1] and len(out['edges'][0]) == 25


# ---- the reference's own case-study inputs (tests/golden/io/case_studies, made by make_case_studies.py) ---------------
CASE_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'io', 'case_studies')


@pytest.mark.parametrize('case,n_frag', [('hsp90', 23), ('jnk', 29)])
def test_case_study_fragments_and_pocket_against_the_io_oracle(case, n_frag):
    """``read_molecule`` / ``get_pocket`` on real inputs (3hz1 / 3fi3 fragments, proteins trimmed to 12 A) against the
    independent restatement in oracle/io_oracle.py (RDKit / Bio.PDB themselves are absent: their parsing stays unpinned)."""
    from difflinker_amd import io
    from oracle import io_oracle
    sdf = os.path.join(CASE_DIR, f'{case}_fragments.sdf')
    pdb = os.path.join(CASE_DIR, f'{case}_protein_12A.pdb')
    mol = io.read_molecule(sdf)
    syms, xyz = io_oracle.heavy_atoms_of_sdf(sdf)
    assert len(mol) == n_frag == len(syms) and list(mol.symbols) == syms
    assert np.array_equal(np.asarray(mol.positions, dtype=np.float64), xyz)
    for bb in (False, True):
        pos, one_hot, charges = io.get_pocket(mol, pdb, backbone_atoms_only=bb)
        want_pos, want_sym = io_oracle.pocket_of_protein(pdb, xyz, backbone_atoms_only=bb)
        assert pos.shape == want_pos.shape and len(want_sym) > 50
        assert np.array_equal(np.asarray(pos, dtype=np.float32), want_pos)
        from difflinker_amd import const
        assert [const.GEOM_IDX2ATOM[int(k)] for k in one_hot.argmax(1)] == want_sym
        assert charges.tolist() == [io_oracle.GEOM_ATOMS[s] for s in want_sym]
    # the trimmed protein still holds residues the 6 A rule rejects
    n_all = sum(1 for ln in open(pdb) if ln.startswith('ATOM'))
    assert len(want_sym) < n_all


def test_save_xyz_file_writes_what_the_reference_writes(tmp_path):
    """Text of the files against ``src/visualizer.py::save_xyz_file`` itself (fixture xyz_writer.npz)."""
    from difflinker_amd import io
    z = np.load(os.path.join(os.path.dirname(CASE_DIR), '..', 'xyz_writer.npz'))
    for tag, is_geom in (('zinc', False), ('geom', True)):
        one_hot, pos, mask = (torch.from_numpy(z[f'{tag}_{k}']) for k in ('one_hot', 'pos', 'mask'))
        io.save_xyz_file(str(tmp_path), one_hot, pos, mask, names=[f'm{i}' for i in range(3)], is_geom=is_geom, suffix='x')
        for i in range(3):
            assert open(tmp_path / f'm{i}_x.xyz').read() == str(z[f'{tag}_text'][i])
