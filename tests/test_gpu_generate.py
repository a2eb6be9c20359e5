"""GPU end-to-end tests of the generation drivers (difflinker_amd/generate.py): fragments file -> size sampler ->
DDPM.sample_chain (HIP) -> .xyz files, with and without a protein pocket (reference generate.py:62-167,
generate_with_protein.py:151-300).  Random-init weights: what is checked is the data flow around the sampler — the
fragment atoms come back at their input coordinates, the requested number of linker atoms is appended, pocket atoms
are left out of the output, a random seed reproduces the files."""
import os

import numpy as np
import pytest
import torch

from helpers import seeded_size_state_dict

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
IO_DIR = os.path.join(HERE, 'golden', 'io')


def ddpm_hparams(pockets):
    hp = dict(in_node_nf=9 if pockets else 8, n_dims=3, context_node_nf=2 if pockets else 1, hidden_nf=128,
              activation='silu', tanh=False, n_layers=2, attention=False, norm_constant=1e-6, inv_sublayers=2,
              sin_embedding=False, normalization_factor=100, aggregation_method='sum', diffusion_steps=500,
              diffusion_noise_schedule='polynomial_2', diffusion_noise_precision=1e-5, diffusion_loss_type='l2',
              normalize_factors=[1, 4, 10], include_charges=False, model='egnn_dynamics', data_path='d',
              train_data_prefix='MOAD_train.full' if pockets else 'zinc_final_train',
              val_data_prefix='MOAD_val.full' if pockets else 'zinc_final_val', batch_size=8, lr=2e-4,
              torch_device='cuda:0', test_epochs=20, n_stability_samples=10, normalization='batch_norm',
              anchors_context=False)
    if pockets:
        hp['graph_type'] = 'FC-10A-4A'
    return hp


def read_xyz(path):
    lines = open(path).read().splitlines()
    n = int(lines[0])
    syms = [ln.split()[0] for ln in lines[2:2 + n]]
    pos = np.array([[float(v) for v in ln.split()[1:4]] for ln in lines[2:2 + n]])
    return syms, pos


def test_generate_from_sdf_fixed_and_uniform_sizes(tmp_path):
    from difflinker_amd import DDPM, io
    from difflinker_amd.generate import generate
    torch.manual_seed(0)
    ddpm = DDPM(**ddpm_hparams(False))
    frag = io.read_molecule(os.path.join(IO_DIR, 'frag.sdf'))
    files = generate(os.path.join(IO_DIR, 'frag.sdf'), ddpm, str(tmp_path / 'a'), n_samples=3, n_steps=5, linker_size='4')
    assert [os.path.basename(f) for f in files] == [f'output_{i}_frag_.xyz' for i in range(3)]
    for f in files:
        syms, pos = read_xyz(f)
        assert len(syms) == len(frag) + 4
        assert syms[:len(frag)] == frag.symbols, 'fragment atom types are carried through the chain'
        assert np.abs(pos[:len(frag)] - frag.positions).max() <= 1e-4, 'fragments return to the input frame'
        assert np.isfinite(pos).all()
    torch.manual_seed(1)
    files = generate(os.path.join(IO_DIR, 'frag.sdf'), ddpm, str(tmp_path / 'b'), n_samples=4, n_steps=5, linker_size='2,6')
    sizes = [len(read_xyz(f)[0]) - len(frag) for f in files]
    assert all(2 <= s <= 6 for s in sizes)


def test_generate_with_size_predictor_checkpoint(tmp_path):
    from difflinker_amd import DDPM, io
    from difflinker_amd.generate import generate
    from difflinker_amd.linker_size import SizeClassifier
    torch.manual_seed(0)
    ddpm = DDPM(**ddpm_hparams(False))
    clf = SizeClassifier(in_node_nf=8, hidden_nf=128, out_node_nf=10, n_layers=3)
    clf.load_state_dict(seeded_size_state_dict(8, 128, 10, 3, seed=4, prefix='gnn.'))
    ckpt = str(tmp_path / 'size.ckpt')
    torch.save({'hyper_parameters': dict(data_path='d', train_data_prefix='t', val_data_prefix='v', in_node_nf=8,
                                         hidden_nf=128, out_node_nf=10, n_layers=3, batch_size=64, lr=1e-3,
                                         torch_device='cpu'),
                'state_dict': clf.state_dict()}, ckpt)
    frag = io.read_molecule(os.path.join(IO_DIR, 'frag.sdf'))
    torch.manual_seed(7)
    files = generate(os.path.join(IO_DIR, 'frag.sdf'), ddpm, str(tmp_path / 'c'), n_samples=5, n_steps=4, linker_size=ckpt)
    sizes = [len(read_xyz(f)[0]) - len(frag) for f in files]
    assert all(3 <= s <= 12 for s in sizes), sizes


def test_generate_with_protein_hides_pocket_and_is_seed_reproducible(tmp_path):
    from difflinker_amd import DDPM, io
    from difflinker_amd.generate import generate_with_protein, generate_with_pocket
    torch.manual_seed(0)
    ddpm = DDPM(**ddpm_hparams(True))
    frag = io.read_molecule(os.path.join(IO_DIR, 'frag.sdf'))
    prot = os.path.join(IO_DIR, 'protein.pdb')
    kw = dict(backbone_atoms_only=False, model=ddpm, n_samples=2, n_steps=4, linker_size='3', random_seed=11)
    f1 = generate_with_protein(os.path.join(IO_DIR, 'frag.sdf'), prot, output_dir=str(tmp_path / 'p1'), **kw)
    f2 = generate_with_protein(os.path.join(IO_DIR, 'frag.sdf'), prot, output_dir=str(tmp_path / 'p2'), **kw)
    for a, b in zip(f1, f2):
        assert open(a).read() == open(b).read(), 'same random_seed -> identical files'
        syms, pos = read_xyz(a)
        assert len(syms) == len(frag) + 3, 'pocket atoms are masked out of the output'
        assert syms[:len(frag)] == frag.symbols
        assert np.abs(pos[:len(frag)] - frag.positions).max() <= 1e-4
    # the pocket-file variant refuses elements outside the vocabulary (the zinc ion), like the reference's lookup
    with pytest.raises(KeyError, match='ZN'):
        generate_with_pocket(os.path.join(IO_DIR, 'frag.sdf'), prot, output_dir=str(tmp_path / 'p3'), **kw)
    f3 = generate_with_pocket(os.path.join(IO_DIR, 'frag.sdf'), prot, output_dir=str(tmp_path / 'p4'),
                              **dict(kw, backbone_atoms_only=True))
    assert len(read_xyz(f3[0])[0]) == len(frag) + 3


def toy_dataset(n_mols, nf, pockets, seed, device):
    g = torch.Generator().manual_seed(seed)
    data = []
    for k in range(n_mols):
        n_frag, n_link, n_pock = 6 + k, 3, (7 if pockets else 0)
        n = n_frag + n_pock + n_link
        frag_only = torch.zeros(n); frag_only[:n_frag] = 1
        pock = torch.zeros(n); pock[n_frag:n_frag + n_pock] = 1
        link = torch.zeros(n); link[n_frag + n_pock:] = 1
        item = {'uuid': k, 'name': f'mol{k}', 'positions': 2.0 * torch.randn((n, 3), generator=g),
                'one_hot': torch.nn.functional.one_hot(torch.randint(0, nf, (n,), generator=g), nf).float(),
                'charges': torch.zeros(n), 'anchors': torch.zeros(n), 'fragment_mask': frag_only + pock,
                'linker_mask': link, 'num_atoms': n}
        if pockets:
            item['fragment_only_mask'] = frag_only
            item['pocket_mask'] = pock
        data.append({k_: (v.to(device) if torch.is_tensor(v) else v) for k_, v in item.items()})
    return data


@pytest.mark.parametrize('pockets', [False, True])
def test_sample_driver_writes_and_resumes(tmp_path, pockets):
    """sample.py's loop on a preprocessed toy set: ground truth / fragments (/ pocket) once, n_samples molecules per
    entry, a second call regenerates nothing."""
    from difflinker_amd import DDPM
    from difflinker_amd.sample import sample
    dev = 'cuda:0'
    prefix = 'MOAD_test.full' if pockets else 'zinc_final_test'
    torch.save(toy_dataset(3, 9 if pockets else 8, pockets, seed=3, device=dev),
               os.path.join(tmp_path, 'MOAD_test_full.pt' if pockets else 'zinc_final_test.pt'))
    torch.manual_seed(0)
    hp = ddpm_hparams(pockets)
    hp.update(batch_size=2, data_path=str(tmp_path))
    ddpm = DDPM(**hp)
    torch.manual_seed(5)
    out = sample(ddpm, str(tmp_path / 'samples'), prefix, n_samples=2, device=dev, n_steps=4)
    for u, n_frag in zip('012', (6, 7, 8)):
        files = sorted(os.listdir(os.path.join(out, u)))
        assert files == (['0_.xyz', '1_.xyz', 'frag_.xyz'] + (['pock_.xyz'] if pockets else []) + ['true_.xyz'])
        assert len(read_xyz(os.path.join(out, u, 'frag_.xyz'))[0]) == n_frag
        assert len(read_xyz(os.path.join(out, u, 'true_.xyz'))[0]) == n_frag + 3
        if pockets:
            assert len(read_xyz(os.path.join(out, u, 'pock_.xyz'))[0]) == 7
        syms, pos = read_xyz(os.path.join(out, u, '1_.xyz'))
        assert len(syms) == n_frag + 3 and np.isfinite(pos).all()
        # fragments are written in the same (centre-of-mass) frame as the ground truth
        fsyms, fpos = read_xyz(os.path.join(out, u, 'frag_.xyz'))
        assert syms[:n_frag] == fsyms and np.abs(pos[:n_frag] - fpos).max() <= 1e-4
    stamp = {u: os.path.getmtime(os.path.join(out, u, '1_.xyz')) for u in '012'}
    sample(ddpm, str(tmp_path / 'samples'), prefix, n_samples=2, device=dev, n_steps=4)
    assert stamp == {u: os.path.getmtime(os.path.join(out, u, '1_.xyz')) for u in '012'}


def test_sample_trajectories_driver(tmp_path):
    from difflinker_amd import DDPM
    from difflinker_amd.sample import sample_trajectories
    dev = 'cuda:0'
    torch.save(toy_dataset(2, 8, False, seed=4, device=dev), os.path.join(tmp_path, 'zinc_final_test.pt'))
    torch.manual_seed(0)
    hp = ddpm_hparams(False)
    hp.update(data_path=str(tmp_path))
    ddpm = DDPM(**hp)
    chains_dir, final_dir = sample_trajectories(ddpm, str(tmp_path / 'chains'), 'zinc_final_test', keep_frames=3,
                                                device=dev, n_steps=6)
    assert sorted(os.listdir(chains_dir)) == ['0', '1']
    assert sorted(os.listdir(os.path.join(chains_dir, '1'))) == ['1_0_.xyz', '1_1_.xyz', '1_2_.xyz']
    assert sorted(os.listdir(final_dir)) == ['0_pred_.xyz', '0_true_.xyz', '1_pred_.xyz', '1_true_.xyz']
    assert len(read_xyz(os.path.join(final_dir, '1_pred_.xyz'))[0]) == 7 + 3
    # frame 0 of the chain is the final sample
    assert open(os.path.join(chains_dir, '1', '1_0_.xyz')).read() == open(os.path.join(final_dir, '1_pred_.xyz')).read()


@pytest.mark.parametrize('case,anchors', [('hsp90', '12,22'), ('jnk', None)])
def test_generate_with_protein_on_the_reference_case_studies(tmp_path, case, anchors):
    """The reference's own inputs (case_studies/hsp90: 3hz1 fragments with anchors 12,22 as in its README;
    case_studies/jnk: 3fi3 fragments), protein trimmed to 12 A around the fragments: the 6 A pocket of several hundred
    atoms (N ~ 350-450, the host-driven pocket chain) goes through ``generate_with_protein`` end to end."""
    from difflinker_amd import DDPM, io
    from difflinker_amd.generate import generate_with_protein
    from oracle import io_oracle
    cdir = os.path.join(IO_DIR, 'case_studies')
    sdf, pdb = os.path.join(cdir, f'{case}_fragments.sdf'), os.path.join(cdir, f'{case}_protein_12A.pdb')
    hp = ddpm_hparams(True)
    if anchors:
        hp.update(anchors_context=True, context_node_nf=3, center_of_mass='anchors')
    torch.manual_seed(0)
    ddpm = DDPM(**hp)
    frag = io.read_molecule(sdf)
    pocket_pos, pocket_sym = io_oracle.pocket_of_protein(pdb, np.asarray(frag.positions, dtype=np.float64))
    assert 150 < len(pocket_sym) < 600
    files = generate_with_protein(sdf, pdb, backbone_atoms_only=False, model=ddpm, output_dir=str(tmp_path / case),
                                  n_samples=3, n_steps=4, linker_size='5', anchors=anchors, random_seed=3)
    assert len(files) == 3
    for f in files:
        syms, pos = read_xyz(f)
        assert len(syms) == len(frag) + 5, 'fragments + linker only: the pocket atoms stay out of the file'
        assert syms[:len(frag)] == list(frag.symbols)
        assert np.abs(pos[:len(frag)] - np.asarray(frag.positions)).max() <= 2e-4, 'fragments return to the input frame'
        assert np.isfinite(pos).all()
