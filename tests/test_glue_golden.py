"""Host glue of the hot path (row a15) against outputs of the UNMODIFIED reference (``tests/golden/ddpm_glue.npz``,
made by ``make_golden.py: ddpm_glue`` from ``src/datasets.py`` / ``src/lightning.py`` imported with stand-ins for the
packages this image lacks).  CPU part: the product's ``collate`` / ``create_templates_for_linker_generation`` and the
oracle restatement of ``DDPM.sample_chain`` (``oracle/ddpm_oracle.py``); the GPU part lives in
``tests/test_gpu_parity_hard.py``."""
import os

import numpy as np
import pytest
import torch

from helpers import GLUE_HPARAMS, glue_cases, glue_molecules, seeded_state_dict, rel_l2
from oracle import ddpm_oracle, edm_oracle
from oracle.egnn_oracle import EGNNConfig


def load(golden_dir):
    z = np.load(os.path.join(golden_dir, 'ddpm_glue.npz'))
    return {k: torch.from_numpy(z[k]) if z[k].shape != () else z[k].item() for k in z.files}


KEYS = ('positions', 'one_hot', 'anchors', 'fragment_mask', 'linker_mask', 'atom_mask', 'edge_mask')


@pytest.mark.parametrize('case', glue_cases(), ids=[c[0] for c in glue_cases()])
def test_collate_and_templates_match_the_reference(golden_dir, case):
    from difflinker_amd.datasets import collate, create_templates_for_linker_generation
    tag, over, pockets, sizes = case
    g = load(golden_dir)
    nf = dict(GLUE_HPARAMS, **over)['in_node_nf']
    mols = glue_molecules(pockets, nf, seed=400 + len(tag))
    for name, fn_c, fn_t in (('product', collate, create_templates_for_linker_generation),
                             ('oracle', ddpm_oracle.collate, ddpm_oracle.create_templates)):
        data = fn_c(mols)
        templ = fn_t(data, torch.tensor(sizes))
        for k in KEYS + (('fragment_only_mask', 'pocket_mask') if pockets else ()):
            for stage, d in (('collate', data), ('template', templ)):
                want = g[f'{tag}.{stage}.{k}']
                assert d[k].dtype == want.dtype and tuple(d[k].shape) == tuple(want.shape), (name, stage, k)
                assert torch.equal(d[k], want), (name, stage, k)
        assert templ['num_atoms'] == [int(data['fragment_mask'][i].sum()) + s for i, s in enumerate(sizes)] or \
            [int(n) for n in templ['num_atoms']] == [int(data['fragment_mask'][i].sum()) + s for i, s in enumerate(sizes)]


@pytest.mark.parametrize('case', glue_cases(), ids=[c[0] for c in glue_cases()])
def test_oracle_sample_chain_glue_matches_the_reference(golden_dir, case):
    """lightning.py:405-463 restated in ``oracle/ddpm_oracle.py``: context assembly (anchors on/off, pockets branch),
    centre-of-mass mask by dataset type, then the oracle sampler — against the reference's chain."""
    tag, over, pockets, sizes = case
    g = load(golden_dir)
    hp = dict(GLUE_HPARAMS, **over)
    nf, ctx, L, T = hp['in_node_nf'], hp['context_node_nf'], hp['n_layers'], g['T']
    graph_type = hp.get('graph_type') or ('4A' if pockets else 'FC')
    sd = seeded_state_dict(nf + ctx + 1, 128, L, 300 + len(tag), coord_gain=0.02)
    cfg = EGNNConfig(in_node_nf=nf, context_node_nf=ctx, n_layers=L, graph_type=graph_type)
    edm = edm_oracle.EDMOracle(edm_oracle.make_dynamics_oracle(sd, cfg), in_node_nf=nf, timesteps=500)
    edm.T = T
    mols = glue_molecules(pockets, nf, seed=400 + len(tag))
    data = ddpm_oracle.collate(mols)
    B, N = g[f'{tag}.template.positions'].shape[:2]
    bank = edm_oracle.NoiseBank.generate(T, B, N, 3, nf, seed=500 + len(tag))
    nx, nh = bank.stacked()
    assert np.allclose([float(nx.double().sum()), float(nh.double().sum())], g[f'{tag}.noise_checksum'].numpy(), rtol=0, atol=1e-9)
    chain, node_mask, _ = ddpm_oracle.sample_chain(edm, data, sizes, bank, keep_frames=2,
                                                   anchors_context=hp['anchors_context'], pockets=pockets,
                                                   moad_dataset=pockets, center_of_mass=hp.get('center_of_mass', 'fragments'))
    want = g[f'{tag}.chain']
    assert torch.equal(node_mask, g[f'{tag}.node_mask'])
    assert chain.shape == want.shape
    err = rel_l2(chain[..., :3], want[..., :3])
    assert err <= 1e-5, err
    assert torch.equal(chain[0][..., 3:], want[0][..., 3:]), 'one-hot atom types'
