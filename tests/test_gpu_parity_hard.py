"""GPU parity, the harder cases ("next round" item 1): the reference's own C1 run, the host glue against the
reference's ``DDPM.sample_chain``, the pocket path at the C4 size and on a T=50 chain with the threshold-sensitive pairs
counted, a T=500 chain whose coordinate head really moves atoms, and the split-fp16 arithmetic judged against fp64."""
import os

import numpy as np
import pytest
import torch

import test_gpu_parity as P
from helpers import GLUE_HPARAMS, glue_cases, glue_molecules, ragged_fc_molecules, seeded_state_dict, rel_l2, max_abs
from oracle import edm_oracle, egnn_oracle
from oracle.egnn_oracle import EGNNConfig

pytestmark = pytest.mark.gpu


def test_c1_chain_vs_reference_golden(golden_dir):
    """BASELINE config C1 (ZINC hparams, 8 blocks, B=8, N=30, edm.T=50) sampled by the unmodified reference
    (tests/golden/make_golden.py: c1_chain) vs the fused HIP chain fed the same noise bank."""
    from difflinker_amd import EDM, synthetic
    from difflinker_amd.datasets import collate
    g = P.load_golden(golden_dir, 'c1_chain')
    nf, L, T = g['nf'], g['n_layers'], g['T']
    inp = synthetic.sampler_inputs(collate(ragged_fc_molecules(g['sizes'].tolist(), g['linkers'].tolist(), nf, seed=g['data_seed'])))
    B, N = inp['x'].shape[:2]
    bank = edm_oracle.NoiseBank.generate(T, B, N, 3, nf, seed=g['noise_seed'])
    nx, nh = bank.stacked()
    assert np.allclose([float(nx.double().sum()), float(nh.double().sum())], g['noise_checksum'].numpy(), rtol=0, atol=1e-9), \
        'the seeded CPU generator does not reproduce the bank the fixture was made with'
    for precision in ('f16x3', 'fp32'):
        dyn, sd, cfg = P.make_dynamics(nf, g['ctx'], L, seed=g['weight_seed'], coord_gain=g['coord_gain'], precision=precision)
        edm = EDM(dyn, in_node_nf=nf, n_dims=3, timesteps=500, noise_schedule='polynomial_2', noise_precision=1e-5,
                  loss_type='l2', norm_values=[1, 4, 10]).to(P.dev())
        edm.T = T
        d = {k: v.to(P.dev()) for k, v in inp.items()}
        got = edm.sample_chain(d['x'], d['h'], d['node_mask'], d['fragment_mask'], d['linker_mask'], d['edge_mask'],
                               d['context'], keep_frames=g['keep_frames'], noise_bank=(nx, nh)).cpu()
        P.check_chain(f'reference C1 chain {precision}', got, g['chain'], inp)


@pytest.mark.parametrize('case', glue_cases(), ids=[c[0] for c in glue_cases()])
def test_ddpm_sample_chain_vs_reference_golden(golden_dir, case):
    """``DDPM.sample_chain`` of the product (templates, context with / without anchors, pockets branch, centre-of-mass
    mask by dataset type, lightning.py:405-463) against the chain the UNMODIFIED reference DDPM sampled from the same
    molecules, weights and noise (tests/golden/make_golden.py: ddpm_glue)."""
    from difflinker_amd import DDPM
    from difflinker_amd.datasets import MOADDataset, collate
    tag, over, pockets, sizes = case
    g = P.load_golden(golden_dir, 'ddpm_glue')
    hp = dict(GLUE_HPARAMS, **over, torch_device='cuda:0')
    nf, ctx, L, T = hp['in_node_nf'], hp['context_node_nf'], hp['n_layers'], g['T']
    m = DDPM(**hp)
    m.edm.dynamics.load_state_dict(seeded_state_dict(nf + ctx + 1, 128, L, 300 + len(tag), coord_gain=0.02), strict=True)
    m = m.to(P.dev()).eval()
    m.edm.T = T
    mols = glue_molecules(pockets, nf, seed=400 + len(tag))
    if pockets:
        m.val_dataset = MOADDataset(data=mols)
    data = {k: (v.to(P.dev()) if torch.is_tensor(v) else v) for k, v in collate(mols).items()}
    want = g[f'{tag}.chain']
    B, N = want.shape[1:3]
    bank = edm_oracle.NoiseBank.generate(T, B, N, 3, nf, seed=500 + len(tag))
    inner = m.edm.sample_chain
    m.edm.sample_chain = lambda **kw: inner(noise_bank=bank.stacked(), **kw)      # same draws as the reference run
    chain, node_mask = m.sample_chain(data, sample_fn=lambda d: torch.tensor(sizes), keep_frames=2)
    chain, node_mask = chain.cpu(), node_mask.cpu()
    assert torch.equal(node_mask, g[f'{tag}.node_mask'])
    lm = g[f'{tag}.template.linker_mask']
    P.check_chain(f'reference DDPM.sample_chain [{tag}]', chain, want, {'linker_mask': lm, 'fragment_mask': g[f'{tag}.template.fragment_mask']})
    assert rel_l2(chain, want) <= P.CHAIN_TOL                     # every atom of every frame, fragments included


@pytest.mark.parametrize('precision', ['f16x3', 'fp32'])
def test_pocket_forward_at_c4_size(precision):
    """BASELINE config C4 geometry at its own size (30 fragment + 250 pocket atoms in a 10 A ball + 6-12 linker atoms,
    N = 292, FC-10A-4A, 6 blocks): one forward of B=4 molecules against the oracle."""
    nf = 9
    dyn, sd, cfg = P.make_pocket_dynamics(nf, 6, seed=131, precision=precision)
    inp, z, t = P.pocket_inputs(batch=4, n_frag=30, n_pocket=250, linker=(6, 12), nf=nf, seed=133)
    assert inp['x'].shape[1] >= 286
    ref = egnn_oracle.dynamics_forward_pockets(sd, cfg, t, z, inp['node_mask'], inp['linker_mask'], inp['edge_mask'], inp['context'])
    out = P.run_hip_forward(dyn, inp, z, t)
    ev, eh = P.report(f'pocket fwd C4 size {precision}', out, ref, z)
    assert ev <= P.FWD_TOLS[precision] and eh <= P.FWD_TOLS[precision]
    assert float((out * (1 - inp['node_mask'].float())).abs().max()) == 0.0


def count_threshold_sensitive_pairs(x, node_mask):
    """Pairs of real atoms of one molecule whose membership in the 4 A / 10 A radius graph depends on HOW the distance is
    evaluated: ``torch.cdist`` (the reference, egnn.py:584,591 — a matmul form beyond 25 rows) vs the direct
    ``sqrt(sum (xi - xj)^2)`` the HIP graph builder uses."""
    flips = 0
    for b in range(x.shape[0]):
        real = node_mask[b].reshape(-1) != 0
        xb = x[b][real]
        d_ref = torch.cdist(xb, xb)
        d_dir = (xb[:, None, :] - xb[None, :, :]).pow(2).sum(-1).sqrt()
        for thr in (4.0, 10.0):
            flips += int(((d_ref <= thr) != (d_dir <= thr)).sum())
    return flips


def pairs_near_a_cutoff(x, node_mask, eps):
    """(pair, frame) count of real-atom pairs whose distance is within `eps` of the 4 A or 10 A cut-off."""
    hits = 0
    for b in range(x.shape[0]):
        real = node_mask[b].reshape(-1) != 0
        xb = x[b][real].double()
        d = (xb[:, None, :] - xb[None, :, :]).pow(2).sum(-1).sqrt()
        for thr in (4.0, 10.0):
            hits += int(((d - thr).abs() < eps).sum())
    return hits


def test_pocket_chain_at_c4_size_with_the_graph_pinned():
    """A T = 50 chain at the C4 geometry and depth (30 fragment + 250 pocket + 6..12 linker atoms, N = 291, FC-10A-4A, 6 blocks,
    B = 2) against the oracle, every frame kept.  The radius graph is rebuilt by different code on the two sides (torch.cdist
    on the host, direct differences on the GPU), so the test first PROVES the graphs cannot differ: along the oracle's
    trajectory no pair of atoms comes within 1e-4 A of a cut-off (asserted: 0 of ~4 M pair-frames for this seed), while the two
    implementations' coordinates agree to ~1e-6 A wherever an atom is near anything - every edge set of the chain is the same
    on both sides, and the frame-by-frame parity below is a statement about the arithmetic alone."""
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    from difflinker_amd import EDM
    nf, T, seed = 9, 50, 151
    dyn, sd, cfg = P.make_pocket_dynamics(nf, 6, seed=seed)
    inp, _, _ = P.pocket_inputs(batch=2, n_frag=30, n_pocket=250, linker=(6, 12), nf=nf, seed=seed + 2)
    B, N = inp['x'].shape[:2]
    assert N >= 286
    edm = EDM(dyn, in_node_nf=nf, n_dims=3, timesteps=500, noise_schedule='polynomial_2', noise_precision=1e-5,
              loss_type='l2', norm_values=[1, 4, 10]).to(P.dev())
    edm.T = T
    bank = edm_oracle.NoiseBank.generate(T, B, N, 3, nf, seed=seed + 4)
    orc = edm_oracle.EDMOracle(edm_oracle.make_dynamics_oracle(sd, cfg), in_node_nf=nf, timesteps=500)
    orc.T = T
    want = orc.sample_chain(inp['x'], inp['h'], inp['node_mask'], inp['fragment_mask'], inp['linker_mask'],
                            inp['edge_mask'], inp['context'], bank, keep_frames=T)
    assert torch.isfinite(want).all()
    near = sum(pairs_near_a_cutoff(want[k][..., :3], inp['node_mask'], 1e-4) for k in range(T))
    flips = sum(count_threshold_sensitive_pairs(want[k][..., :3], inp['node_mask']) for k in range(T))
    assert near == 0 and flips == 0, f'{near} pair-frames within 1e-4 A of a cut-off, {flips} formula-sensitive: pick another seed'
    g = {k: v.to(P.dev()) for k, v in inp.items()}
    got = edm.sample_chain(g['x'], g['h'], g['node_mask'], g['fragment_mask'], g['linker_mask'], g['edge_mask'],
                           g['context'], keep_frames=T, noise_bank=bank.stacked()).cpu()
    P.check_chain(f'pocket chain at C4 size, T={T}, graph pinned', got, want, inp)
    # wherever an atom sits near anything the two sides agree far below the 1e-4 A margin the graph proof needs
    close = (want[..., :3].abs().amax(-1) < 50.0) & (inp['node_mask'].squeeze(-1) != 0)[None]
    worst = float(((got[..., :3] - want[..., :3]).norm(dim=-1) * close).max())
    print(f'   largest coordinate difference among atoms within 50 A of the origin, any frame: {worst:.2e} A')
    assert worst < 2e-5


def test_pocket_chain_T50_reports_edge_flips():
    """A 50-step pocket chain (linker atoms move across the 10 A cut-off of ~100 pocket atoms) against the oracle, every
    frame kept; pairs whose graph membership depends on the distance formula are counted along the oracle's trajectory."""
    from difflinker_amd import EDM
    nf, T = 9, 50
    dyn, sd, cfg = P.make_pocket_dynamics(nf, 2, seed=135)
    inp, _, _ = P.pocket_inputs(batch=2, n_frag=12, n_pocket=100, linker=(5, 8), nf=nf, seed=137)
    B, N = inp['x'].shape[:2]
    edm = EDM(dyn, in_node_nf=nf, n_dims=3, timesteps=500, noise_schedule='polynomial_2', noise_precision=1e-5,
              loss_type='l2', norm_values=[1, 4, 10]).to(P.dev())
    edm.T = T
    bank = edm_oracle.NoiseBank.generate(T, B, N, 3, nf, seed=139)
    orc = edm_oracle.EDMOracle(edm_oracle.make_dynamics_oracle(sd, cfg), in_node_nf=nf, timesteps=500)
    orc.T = T
    want = orc.sample_chain(inp['x'], inp['h'], inp['node_mask'], inp['fragment_mask'], inp['linker_mask'],
                            inp['edge_mask'], inp['context'], bank, keep_frames=T)
    g = {k: v.to(P.dev()) for k, v in inp.items()}
    got = edm.sample_chain(g['x'], g['h'], g['node_mask'], g['fragment_mask'], g['linker_mask'], g['edge_mask'],
                           g['context'], keep_frames=T, noise_bank=bank.stacked()).cpu()
    flips = sum(count_threshold_sensitive_pairs(want[k][..., :3], inp['node_mask']) for k in range(T))
    moved = float(((want[0][..., :3] - want[T - 1][..., :3]) * inp['linker_mask']).norm(dim=-1).max())
    print(f'[pocket chain T={T}] formula-sensitive pairs along the chain: {flips}; largest linker displacement {moved:.2f} A')
    P.check_chain(f'pocket chain T={T}', got, want, inp)


def test_chain_T500_with_a_live_coordinate_head():
    """The full 500-step chain on 8 molecules with a coordinate head that moves atoms (xavier gain 0.02, twenty times the
    reference's init; the oracle stays finite with linker coordinates of several hundred A), against the oracle.  The
    head's influence on the final coordinates is measured (same chain with the head zeroed) and must exceed the parity
    tolerance a hundredfold, so a wrong velocity could not hide."""
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    from difflinker_amd import EDM, synthetic
    from difflinker_amd.datasets import collate
    nf, L, T, gain = 9, 6, 500, 0.02
    sizes, linkers = [30, 24, 27, 29, 25, 30, 26, 28], [5, 3, 8, 4, 6, 7, 3, 5]
    dyn, sd, cfg = P.make_dynamics(nf, 1, L, seed=95, coord_gain=gain)
    inp = synthetic.sampler_inputs(collate(ragged_fc_molecules(sizes, linkers, nf, seed=91)))
    B, N = inp['x'].shape[:2]
    edm = EDM(dyn, in_node_nf=nf, n_dims=3, timesteps=500, noise_schedule='polynomial_2', noise_precision=1e-5,
              loss_type='l2', norm_values=[1, 4, 10]).to(P.dev())
    bank = edm_oracle.NoiseBank.generate(T, B, N, 3, nf, seed=92)
    orc = edm_oracle.EDMOracle(edm_oracle.make_dynamics_oracle(sd, cfg), in_node_nf=nf, timesteps=500)
    want = orc.sample_chain(inp['x'], inp['h'], inp['node_mask'], inp['fragment_mask'], inp['linker_mask'],
                            inp['edge_mask'], inp['context'], bank, keep_frames=5)
    assert torch.isfinite(want).all()
    # the head matters: the same chain with the head switched off ends somewhere else entirely
    sd0 = {k: (torch.zeros_like(v) if 'coord_mlp.4' in k else v) for k, v in sd.items()}
    orc0 = edm_oracle.EDMOracle(edm_oracle.make_dynamics_oracle(sd0, cfg), in_node_nf=nf, timesteps=500)
    bank.reset()
    base = orc0.sample_chain(inp['x'], inp['h'], inp['node_mask'], inp['fragment_mask'], inp['linker_mask'],
                             inp['edge_mask'], inp['context'], bank, keep_frames=1)
    lm = inp['linker_mask']
    effect = rel_l2(want[0][..., :3] * lm, base[0][..., :3] * lm)
    print(f'[T=500 live head] effect of the coordinate head on the final linker coordinates: rel-L2 {effect:.3f}')
    assert effect > 100 * P.CHAIN_TOL
    g = {k: v.to(P.dev()) for k, v in inp.items()}
    got = edm.sample_chain(g['x'], g['h'], g['node_mask'], g['fragment_mask'], g['linker_mask'], g['edge_mask'],
                           g['context'], keep_frames=5, noise_bank=bank.stacked()).cpu()
    P.check_chain('chain T=500, live coordinate head', got, want, inp)


def test_chain_T500_with_a_live_coordinate_head_geom_sized():
    """The same at the benchmark's molecule size (the live-head chain stopped at 30 atoms): four molecules
    of 41..50 atoms, 6 blocks, T = 500, coordinate head at xavier gain 0.02 - the oracle stays finite (largest coordinate
    ~ 760 A) - on one compute unit per molecule and on teams."""
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    from difflinker_amd import EDM, synthetic
    from difflinker_amd.datasets import collate
    nf, L, T, gain = 9, 6, 500, 0.02
    sizes, linkers = [50, 44, 41, 47], [8, 6, 5, 9]
    dyn, sd, cfg = P.make_dynamics(nf, 1, L, seed=96, coord_gain=gain)
    inp = synthetic.sampler_inputs(collate(ragged_fc_molecules(sizes, linkers, nf, seed=93)))
    B, N = inp['x'].shape[:2]
    edm = EDM(dyn, in_node_nf=nf, n_dims=3, timesteps=500, noise_schedule='polynomial_2', noise_precision=1e-5,
              loss_type='l2', norm_values=[1, 4, 10]).to(P.dev())
    bank = edm_oracle.NoiseBank.generate(T, B, N, 3, nf, seed=94)
    orc = edm_oracle.EDMOracle(edm_oracle.make_dynamics_oracle(sd, cfg), in_node_nf=nf, timesteps=500)
    want = orc.sample_chain(inp['x'], inp['h'], inp['node_mask'], inp['fragment_mask'], inp['linker_mask'],
                            inp['edge_mask'], inp['context'], bank, keep_frames=1)
    assert torch.isfinite(want).all()
    lm = inp['linker_mask']
    moved = float(((want[0][..., :3] - inp['x']) * lm).norm(dim=-1).max())
    print(f'[T=500 live head, 41..50 atoms] largest linker displacement {moved:.1f} A')
    assert moved > 10.0
    g = {k: v.to(P.dev()) for k, v in inp.items()}
    for team in (1, 'auto'):
        dyn.team = team
        got = edm.sample_chain(g['x'], g['h'], g['node_mask'], g['fragment_mask'], g['linker_mask'], g['edge_mask'],
                               g['context'], keep_frames=1, noise_bank=bank.stacked()).cpu()
        P.check_chain(f'chain T=500, live coordinate head, 41..50 atoms, team={team}', got, want, inp)


def test_headline_arithmetic_against_the_exact_mode_at_full_size():
    """The benchmark's own launch - config C2, B = 256, T = 500, in-kernel noise - once in the default f16x3 arithmetic and once
    in the exact-fp32 MFMA mode: the two final samples agree to 1e-5 on the linker coordinates and in every atom type, which
    ties the headline's arithmetic to the exact one at full size (the oracle cannot follow there: an hour per chain)."""
    from difflinker_amd import Dynamics, EDM, synthetic
    data, cfg = synthetic.make_batch('C2', seed=1000)
    inp = {k: v.to(P.dev()) for k, v in synthetic.sampler_inputs(data).items()}
    chains = {}
    for precision in ('f16x3', 'fp32'):
        torch.manual_seed(0)
        dyn = Dynamics(n_dims=3, in_node_nf=cfg['nf'], context_node_nf=cfg['ctx'], hidden_nf=128, n_layers=cfg['n_layers'],
                       norm_constant=1e-6, normalization='batch_norm')
        dyn.precision = precision
        edm = EDM(dyn, in_node_nf=cfg['nf'], n_dims=3, timesteps=500, noise_schedule='polynomial_2', noise_precision=1e-5,
                  loss_type='l2', norm_values=[1, 4, 10]).to(P.dev())
        edm.noise_source, edm.noise_seed = 'philox', 5
        chains[precision] = edm.sample_chain(keep_frames=1, **inp).cpu()
    lm = inp['linker_mask'].cpu()
    a, b = chains['f16x3'][0], chains['fp32'][0]
    ex = rel_l2(a[..., :3] * lm, b[..., :3] * lm)
    mism = int((a[..., 3:] != b[..., 3:]).any(-1).sum())
    print(f'[C2 B=256 T=500, f16x3 vs fp32 mode] final linker-x rel-L2 {ex:.3e}, atom-type mismatches {mism}')
    assert torch.isfinite(a).all() and ex <= 1e-5 and mism == 0


def test_split_fp16_is_fp32_class_against_the_fp64_oracle():
    """f16x3 (default) and exact-fp32 arithmetic judged against the fp64 oracle on a C2-shaped batch (B=64, N=50,
    6 blocks): the split scheme must not be worse than twice the fp32 mode's own rounding error."""
    from difflinker_amd import synthetic
    nf, L = 9, 6
    data, _ = synthetic.make_batch('C2', seed=1, batch=64)
    inp = synthetic.sampler_inputs(data)
    B, N = inp['x'].shape[:2]
    g = torch.Generator().manual_seed(4)
    z = torch.cat([inp['x'], inp['h']], dim=2) * inp['fragment_mask'] + torch.randn((B, N, 3 + nf), generator=g) * inp['linker_mask']
    t = torch.full((B, 1), 0.37)
    sd = seeded_state_dict(nf + 2, 128, L, 80, coord_gain=0.02)
    cfg = EGNNConfig(in_node_nf=nf, context_node_nf=1, n_layers=L)
    sd64 = {k: v.double() for k, v in sd.items()}
    ref = egnn_oracle.dynamics_forward(sd64, cfg, t.double(), z.double(), inp['node_mask'], inp['linker_mask'].double(),
                                       inp['edge_mask'], inp['context'].double())
    err = {}
    for precision in ('fp32', 'f16x3'):
        dyn, _, _ = P.make_dynamics(nf, 1, L, seed=80, precision=precision)
        out = P.run_hip_forward(dyn, inp, z, t).double()
        err[precision] = (rel_l2(out[..., 3:], ref[..., 3:]), float((out[..., :3] - ref[..., :3]).norm()))
    print(f'[vs fp64] h rel-L2: fp32 {err["fp32"][0]:.3e}, f16x3 {err["f16x3"][0]:.3e}; '
          f'vel abs-L2: fp32 {err["fp32"][1]:.3e}, f16x3 {err["f16x3"][1]:.3e}')
    assert err['f16x3'][0] <= 2.0 * err['fp32'][0]
    assert err['f16x3'][1] <= 2.0 * err['fp32'][1]


def _overflowing_head_case(special_is_linker, sizes=(20, 12), linkers=(5, 4)):
    """One block whose coordinate head overflows for ONE receiving atom only: every GCL weight is zero (h stays the embedding),
    the embedding gives atom type 7 - one atom of molecule 0 - the feature h[0] = 1e10, the head reads the receiver's h[0] and
    multiplies it up to 1e42.  That receiver's sum of trans = coord_diff * inf holds inf and, from its own diagonal edge
    (coord_diff = 0, edge mask 0), NaN (egnn.py:106-112)."""
    nf = 9
    inp, z, t = P.ragged_inputs(list(sizes), list(linkers), nf, seed=300)
    special = sizes[0] - 3 if special_is_linker else 3        # molecule 0: its last linkers[0] atoms are the linker
    z[:, :, 3 + 7] = 0.0
    z[0, special, 3 + 7] = 1.0
    sd = seeded_state_dict(nf + 2, 128, 1, 301)
    for v in sd.values():
        v.zero_()
    # (every single weight and intermediate stays below 1e22 - the range of the f16x3 scales, DESIGN 'limits' - only the head's
    # last product, 1e10 * 1e32, leaves fp32)
    sd['dynamics.embedding.weight'][0, 7] = 1e10
    sd['dynamics.e_block_0.gcl_equiv.coord_mlp.0.weight'][0, 0] = 1e10
    sd['dynamics.e_block_0.gcl_equiv.coord_mlp.2.weight'][0, 0] = 1e12
    sd['dynamics.e_block_0.gcl_equiv.coord_mlp.4.weight'][0, 0] = 1e10
    return inp, z, t, sd, EGNNConfig(in_node_nf=nf, context_node_nf=1, n_layers=1), special


@pytest.mark.parametrize('precision', ['f16x3', 'fp32'])
def test_overflow_in_a_linker_atoms_coordinate_sum_raises_like_the_reference(precision):
    from difflinker_amd import Dynamics
    from difflinker_amd.utils import FoundNaNException
    inp, z, t, sd, cfg, _ = _overflowing_head_case(special_is_linker=True)
    with pytest.raises(egnn_oracle.OracleNaN):
        egnn_oracle.dynamics_forward(sd, cfg, t, z, inp['node_mask'], inp['linker_mask'], inp['edge_mask'], inp['context'])
    dyn = Dynamics(n_dims=3, in_node_nf=9, context_node_nf=1, hidden_nf=128, n_layers=1, norm_constant=1e-6)
    dyn.precision = precision
    dyn.load_state_dict(sd, strict=True)
    with pytest.raises(FoundNaNException) as info:
        P.run_hip_forward(dyn.to(P.dev()), inp, z, t)
    assert info.value.only_x_nan_idx == {0} and not info.value.x_h_nan_idx and not info.value.only_h_nan_idx    # molecule 0, coordinates (utils.py:283-289)


@pytest.mark.parametrize('sizes,linkers,team', [((20, 12), (5, 4), '1'), ((20, 12), (5, 4), 'auto'), ((70, 12), (8, 4), 'auto'),
                                               ((120, 12), (9, 4), 'auto')])
@pytest.mark.parametrize('precision', ['f16x3', 'fp32'])
def test_overflow_confined_to_a_fragment_atoms_coordinate_sum_raises_like_the_reference(precision, sizes, linkers, team):
    """The reference sums trans for EVERY receiving atom and multiplies the sum by the linker mask afterwards (egnn.py:110-116):
    an inf / NaN in a FRAGMENT atom's sum becomes NaN * 0 = NaN in its coordinates and Dynamics.forward raises
    FoundNaNException (egnn.py:441-442; generate.py:154-161 then re-samples the batch).  The HIP path skips the sums the mask
    discards - until round 4 unconditionally, a divergence on record.  Round 5: only while a bound of the coordinate head's
    output PROVES them finite (equiv_pass2 / pk_edge_kernel<EQUIV>); this model's bound is 1e42, so every sum is formed and
    multiplied by the mask, and the exception - with the reference's index sets - is raised on one compute unit per molecule,
    on teams (20 and 70 atoms) and on the HBM-resident kernels (120 atoms)."""
    from difflinker_amd import Dynamics
    from difflinker_amd.utils import FoundNaNException
    inp, z, t, sd, cfg, special = _overflowing_head_case(special_is_linker=False, sizes=sizes, linkers=linkers)
    with pytest.raises(egnn_oracle.OracleNaN) as ref:
        egnn_oracle.dynamics_forward(sd, cfg, t, z, inp['node_mask'], inp['linker_mask'], inp['edge_mask'], inp['context'])
    assert ref.value.only_x_nan_idx == {0} and not ref.value.x_h_nan_idx and not ref.value.only_h_nan_idx
    dyn = Dynamics(n_dims=3, in_node_nf=9, context_node_nf=1, hidden_nf=128, n_layers=1, norm_constant=1e-6)
    dyn.precision = precision
    dyn.team = team if team == 'auto' else int(team)
    dyn.load_state_dict(sd, strict=True)
    with pytest.raises(FoundNaNException) as info:
        P.run_hip_forward(dyn.to(P.dev()), inp, z, t)
    assert info.value.only_x_nan_idx == {0} and not info.value.x_h_nan_idx and not info.value.only_h_nan_idx


def test_skipped_coordinate_sums_stay_skipped_for_ordinary_models():
    """...and the proof holds for ordinary weights: the coordinate pass of a seeded-random model (and of the trained-like one)
    still runs over the linker receivers only - same bits as before this check existed is not observable from outside, so the
    test pins the OBSERVABLE side: a forward whose fragment atoms carry huge but finite features is finite and equals the oracle."""
    nf = 9
    dyn, sd, cfg = P.make_dynamics(nf, 1, 2, seed=411)
    inp, z, t = P.ragged_inputs([30, 44], [6, 9], nf, seed=412)
    ref = egnn_oracle.dynamics_forward(sd, cfg, t, z, inp['node_mask'], inp['linker_mask'], inp['edge_mask'], inp['context'])
    out = P.run_hip_forward(dyn, inp, z, t)
    ev, eh = P.report('ordinary model: coordinate pass over linker receivers', out, ref, z)
    assert ev <= P.FWD_TOLS['f16x3'] and eh <= P.FWD_TOLS['f16x3']
