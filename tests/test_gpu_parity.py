"""GPU parity tests: the HIP path (through the Python boundary -> C ABI) against the CPU oracle on the
same seeded inputs, against the golden fixtures of the unmodified reference, and through
size-independent properties (padding invariance, E(3) equivariance, determinism, mask semantics).

Tolerances: one ``Dynamics.forward`` rel-L2 <= 1e-5 (exact-fp32 mode) / <= 2e-6 (f16x3, the default) on vel and h; a full
``sample_chain`` with a shared noise bank rel-L2 <= 1e-4 on the linker coordinates and exact one-hot
atom types (BASELINE.json north_star: 1e-4).
"""
import math
import os

import numpy as np
import pytest
import torch

from helpers import seeded_state_dict, rel_l2, max_abs
from oracle import edm_oracle, egnn_oracle
from oracle.egnn_oracle import EGNNConfig

pytestmark = pytest.mark.gpu

# one forward, rel-L2 on the node features and on the updated coordinates x + vel (see `report`): exact-fp32 MFMA mode 1e-5;
# scaled split-fp16 (f16x3, the default) 2e-6 - its measured class is 2..9e-7 in every case of the suite (profiles/r04/
# pytest_gpu_measured_errors.log), so a 10x regression fails (round 4 allowed 2e-5: VERDICT).  The RAW velocity error is bounded
# by 1e-4 in every forward test (it is the fp32 floor of x_final - x when the update is tiny: up to 4e-5 with the 0.02-gain heads
# used here) and by the forward tolerance itself where the update is of the order of the coordinates
# (test_forward_velocity_with_a_live_coordinate_head: measured 9e-7).
FWD_TOLS = {'fp32': 1e-5, 'f16x3': 2e-6}
FWD_TOL = FWD_TOLS[os.environ.get('DIFFLINKER_PRECISION', 'f16x3')]
CHAIN_TOL = 1e-4


def dev():
    assert torch.cuda.is_available(), 'GPU tests need an MI355X'
    return torch.device('cuda:0')


@pytest.fixture(params=['1', 'auto'], autouse=True)
def compute_units_per_molecule(request, monkeypatch):
    """Every case of this file runs twice: on one compute unit per molecule (the kernels of batches >= 256, i.e. of the
    benchmark) and with Dynamics.team = 'auto' (teams of 8 workgroups per molecule at these small batches)."""
    monkeypatch.setenv('DIFFLINKER_TEAM', request.param)


def make_dynamics(nf, ctx, n_layers, seed, coord_gain=0.02, precision=None):
    from difflinker_amd import Dynamics
    dyn = Dynamics(n_dims=3, in_node_nf=nf, context_node_nf=ctx, hidden_nf=128, n_layers=n_layers,
                   norm_constant=1e-6, normalization='batch_norm')
    if precision is not None:
        dyn.precision = precision
    sd = seeded_state_dict(nf + ctx + 1, 128, n_layers, seed, coord_gain=coord_gain)
    dyn.load_state_dict(sd, strict=True)
    return dyn.to(dev()), sd, EGNNConfig(in_node_nf=nf, context_node_nf=ctx, n_layers=n_layers)


def ragged_inputs(sizes, linkers, nf, seed, ctx=1, n_pad=None):
    from difflinker_amd import synthetic
    from difflinker_amd.datasets import collate
    g = torch.Generator().manual_seed(seed)
    mols = []
    for n, nl in zip(sizes, linkers):
        frag = torch.zeros(n)
        frag[:n - nl] = 1
        types = torch.randint(0, nf, (n,), generator=g)
        mols.append({'positions': 2.0 * torch.randn((n, 3), generator=g),
                     'one_hot': torch.nn.functional.one_hot(types, nf).float(),
                     'anchors': torch.zeros(n), 'fragment_mask': frag, 'linker_mask': 1 - frag, 'num_atoms': n})
    if n_pad is not None and n_pad > max(sizes):
        # a dummy molecule that only forces the padded width; dropped below
        mols.append({'positions': torch.zeros(n_pad, 3), 'one_hot': torch.zeros(n_pad, nf), 'anchors': torch.zeros(n_pad),
                     'fragment_mask': torch.ones(n_pad), 'linker_mask': torch.zeros(n_pad), 'num_atoms': n_pad})
    data = collate(mols)
    if n_pad is not None and n_pad > max(sizes):
        B = len(sizes)
        N = n_pad
        em = data['edge_mask'].view(B + 1, N * N)[:B].reshape(-1, 1)
        data = {k: (v[:B] if torch.is_tensor(v) and v.dim() >= 1 and v.shape[0] == B + 1 else v) for k, v in data.items()}
        data['edge_mask'] = em
    inp = synthetic.sampler_inputs(data)
    if ctx == 2:                                     # e.g. anchors + fragment mask
        inp['context'] = torch.cat([data['anchors'], inp['context']], dim=-1)
    B, N = inp['x'].shape[:2]
    z = torch.cat([inp['x'], inp['h']], dim=2) * inp['fragment_mask'] + \
        torch.randn((B, N, 3 + nf), generator=g) * inp['linker_mask']
    t = torch.rand((B, 1), generator=g)
    return inp, z, t


def run_hip_forward(dyn, inp, z, t, linker_mask='given', edge_mask='given'):
    d = dev()
    lm = inp['linker_mask'].to(d) if linker_mask == 'given' else None
    em = inp['edge_mask'].to(d) if edge_mask == 'given' else edge_mask
    out = dyn.forward(t=t.to(d), xh=z.to(d), node_mask=inp['node_mask'].to(d), linker_mask=lm,
                      edge_mask=em, context=inp['context'].to(d))
    torch.cuda.synchronize()
    return out.cpu()


def report(tag, out, ref, xin=None):
    """rel-L2 of the coordinate and feature parts of one forward.  The coordinate figure is taken on what the sampler consumes,
    ``x_final = x + vel`` (the figure of rounds 2-5 subtracted a ``4 ulp(|x|)`` floor from the velocity error
    before normalising and printed 0 on almost every line - it never bound); the RAW velocity error - a difference of fp32
    coordinates, a few ulp(|x|) of absolute error on both sides, large relative to a tiny update - is printed and bounded by the
    north-star bar 1e-4 in every test, and by the forward tolerance itself where the update is of the order of the coordinates
    (test_forward_velocity_with_a_live_coordinate_head)."""
    raw = rel_l2(out[..., :3], ref[..., :3])
    if xin is not None:
        x0 = xin[..., :3].to(out.dtype)              # (padding rows of the state are zero: z = [x, h] * fragment_mask + noise * linker_mask)
        ev = rel_l2(x0 + out[..., :3], x0 + ref[..., :3])
    else:
        ev = raw
    eh = rel_l2(out[..., 3:], ref[..., 3:])
    print(f'[{tag}] rel-L2 x+vel {ev:.3e} (raw vel {raw:.3e}) h {eh:.3e} | max-abs {max_abs(out, ref):.3e}')
    assert raw <= 1e-4, f'{tag}: raw velocity rel-L2 {raw:.3e} above the 1e-4 bar'
    return ev, eh


def load_golden(golden_dir, name):
    z = np.load(os.path.join(golden_dir, name + '.npz'))
    return {k: torch.from_numpy(z[k]) if z[k].shape != () else z[k].item() for k in z.files}


# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('sizes,linkers,n_layers', [
    ([5], [2], 1),                       # one partial tile, rows of several atoms per tile (generic reduce)
    ([33], [5], 1),                      # two M-tiles, two-atoms-per-tile fast reduce
    ([14, 9, 12, 5], [4, 3, 5, 2], 2),
    ([55, 32, 31, 2, 40], [6, 3, 4, 1, 12], 2),   # LDS limit, tile boundaries, one fragment + one linker atom
    ([50, 35, 44], [8, 3, 12], 6),       # GEOM-sized, full depth
])
@pytest.mark.parametrize('precision', ['f16x3', 'fp32'])
def test_forward_vs_oracle(sizes, linkers, n_layers, precision):
    nf, ctx = 9, 1
    dyn, sd, cfg = make_dynamics(nf, ctx, n_layers, seed=100 + n_layers, precision=precision)
    inp, z, t = ragged_inputs(sizes, linkers, nf, seed=sum(sizes))
    ref = egnn_oracle.dynamics_forward(sd, cfg, t, z, inp['node_mask'], inp['linker_mask'], inp['edge_mask'],
                                       inp['context'])
    out = run_hip_forward(dyn, inp, z, t)
    ev, eh = report(f'fwd sizes={sizes} L={n_layers} {precision}', out, ref, z)
    nm = inp['node_mask'].float()
    assert float((out * (1 - nm)).abs().max()) == 0.0, 'padded rows must be exactly zero'
    assert ev <= FWD_TOLS[precision] and eh <= FWD_TOLS[precision]


@pytest.mark.parametrize('precision', ['f16x3', 'fp32'])
def test_forward_velocity_with_a_live_coordinate_head(precision):
    """One forward with a coordinate head at xavier gain 1.0 (a thousand times the reference's init): the velocity is of the
    order of the coordinates themselves, the ulp(|x|) floor of `report` is irrelevant, and the RAW velocity error must meet
    the forward tolerance - a wrong coordinate update could not hide behind the floor."""
    nf = 9
    dyn, sd, cfg = make_dynamics(nf, 1, 2, seed=107, coord_gain=1.0, precision=precision)
    inp, z, t = ragged_inputs([50, 35, 44, 12], [8, 3, 12, 4], nf, seed=108)
    ref = egnn_oracle.dynamics_forward(sd, cfg, t, z, inp['node_mask'], inp['linker_mask'], inp['edge_mask'], inp['context'])
    out = run_hip_forward(dyn, inp, z, t)
    report(f'fwd, coordinate head gain 1.0, {precision}', out, ref, z)
    lm = inp['linker_mask']
    vel_scale = float((ref[..., :3] * lm).norm() / (z[..., :3] * lm).norm())
    raw = rel_l2(out[..., :3], ref[..., :3])
    print(f'   |velocity| / |x| over the linker atoms: {vel_scale:.3f}; raw velocity rel-L2 {raw:.3e}')
    assert vel_scale > 1e-2, 'the update must dominate the rounding floor for this test to mean anything'
    assert raw <= FWD_TOLS[precision]


def test_forward_vs_reference_golden(golden_dir):
    g = load_golden(golden_dir, 'fc_forward')
    dyn, sd, cfg = make_dynamics(g['nf'], g['ctx'], g['n_layers'], seed=g['weight_seed'], coord_gain=g['coord_gain'])
    inp = {k: g[k] for k in ('node_mask', 'linker_mask', 'edge_mask', 'context')}
    out = run_hip_forward(dyn, inp, g['xh'], g['t'])
    ev, eh = report('golden fc_forward', out, g['out'], g['xh'])
    assert ev <= FWD_TOL and eh <= FWD_TOL
    # scalar-t branch (egnn.py:397-399) on the un-padded molecule 1
    B, N = g['xh'].shape[:2]
    inp1 = {'node_mask': g['node_mask'][1:2, :9], 'linker_mask': g['linker_mask'][1:2, :9],
            'edge_mask': g['edge_mask'].view(B, N, N)[1, :9, :9].reshape(-1, 1), 'context': g['context'][1:2, :9]}
    out1 = run_hip_forward(dyn, inp1, g['xh'][1:2, :9], g['t'][1:2])
    ev, eh = report('golden fc_forward mol1 unpadded', out1, g['out_mol1_unpadded'], g['xh'][1:2, :9])
    assert ev <= FWD_TOL and eh <= FWD_TOL


@pytest.mark.parametrize('nf,ctx', [(8, 1), (9, 2)])
def test_forward_feature_widths(nf, ctx):
    dyn, sd, cfg = make_dynamics(nf, ctx, 2, seed=7)
    inp, z, t = ragged_inputs([20, 17], [5, 4], nf, seed=3, ctx=ctx)
    ref = egnn_oracle.dynamics_forward(sd, cfg, t, z, inp['node_mask'], inp['linker_mask'], inp['edge_mask'],
                                       inp['context'])
    ev, eh = report(f'fwd nf={nf} ctx={ctx}', run_hip_forward(dyn, inp, z, t), ref, z)
    assert ev <= FWD_TOL and eh <= FWD_TOL


def test_forward_without_linker_mask_and_wide_padding():
    nf = 9
    dyn, sd, cfg = make_dynamics(nf, 1, 2, seed=8)
    inp, z, t = ragged_inputs([10, 22], [3, 6], nf, seed=5, n_pad=70)       # N=70 > 64: multi-chunk compaction
    ref = egnn_oracle.dynamics_forward(sd, cfg, t, z, inp['node_mask'], None, inp['edge_mask'], inp['context'])
    ev, eh = report('fwd linker_mask=None N=70', run_hip_forward(dyn, inp, z, t, linker_mask=None), ref, z)
    assert ev <= FWD_TOL and eh <= FWD_TOL


def test_padding_invariance_is_exact():
    """A molecule gives bitwise the same eps_hat alone/un-padded and inside a padded batch."""
    nf = 9
    dyn, sd, cfg = make_dynamics(nf, 1, 2, seed=9)
    inp, z, t = ragged_inputs([30, 12, 21], [5, 4, 6], nf, seed=6)
    full = run_hip_forward(dyn, inp, z, t)
    B, N = z.shape[:2]
    for b, n in enumerate([30, 12, 21]):
        sub = {'node_mask': inp['node_mask'][b:b + 1, :n], 'linker_mask': inp['linker_mask'][b:b + 1, :n],
               'edge_mask': inp['edge_mask'].view(B, N, N)[b, :n, :n].reshape(-1, 1),
               'context': inp['context'][b:b + 1, :n]}
        alone = run_hip_forward(dyn, sub, z[b:b + 1, :n], t[b:b + 1])
        assert torch.equal(alone[0], full[b, :n]), f'molecule {b}'


def test_determinism_bitwise():
    nf = 9
    dyn, sd, cfg = make_dynamics(nf, 1, 3, seed=10)
    inp, z, t = ragged_inputs([50, 37, 44, 9], [8, 3, 12, 2], nf, seed=7)
    a = run_hip_forward(dyn, inp, z, t)
    for _ in range(3):
        assert torch.equal(run_hip_forward(dyn, inp, z, t), a)


def test_rotation_equivariance_and_translation():
    """vel rotates with x, h_final is invariant (E(3) property of egnn.py:295-301,101-117)."""
    nf = 9
    dyn, sd, cfg = make_dynamics(nf, 1, 3, seed=11)
    inp, z, t = ragged_inputs([28, 41], [6, 9], nf, seed=8)
    q, _ = torch.linalg.qr(torch.randn(3, 3, generator=torch.Generator().manual_seed(1)))
    if torch.det(q) < 0:
        q[:, 0] = -q[:, 0]
    zr = z.clone()
    zr[..., :3] = z[..., :3] @ q.T
    a, b = run_hip_forward(dyn, inp, z, t), run_hip_forward(dyn, inp, zr, t)
    e_vel = rel_l2(b[..., :3], a[..., :3] @ q.T)
    e_h = rel_l2(b[..., 3:], a[..., 3:])
    print(f'[equivariance] vel {e_vel:.3e} h {e_h:.3e}')
    assert e_vel <= 1e-4 and e_h <= 1e-4
    zt = z.clone()
    zt[..., :3] = (z[..., :3] + torch.tensor([1.5, -2.0, 0.75])) * inp['node_mask'].float()
    c = run_hip_forward(dyn, inp, zt, t)
    assert rel_l2(c, a) <= 1e-4


def test_edge_mask_sign_convention_is_observable():
    """The int8 {0,-1,-2} mask multiplies messages as-is; a boolean {0,1} off-diagonal mask gives a
    different answer on both the oracle and the HIP path."""
    nf = 9
    dyn, sd, cfg = make_dynamics(nf, 1, 2, seed=12)
    inp, z, t = ragged_inputs([18, 25], [4, 6], nf, seed=9)
    B, N = z.shape[:2]
    nm = inp['node_mask'].view(B, N).to(torch.int8)
    bool_mask = (nm[:, None, :] * nm[:, :, None]) * (1 - torch.eye(N, dtype=torch.int8))
    bool_mask = bool_mask.view(-1, 1)
    ref_i8 = egnn_oracle.dynamics_forward(sd, cfg, t, z, inp['node_mask'], inp['linker_mask'], inp['edge_mask'], inp['context'])
    ref_bool = egnn_oracle.dynamics_forward(sd, cfg, t, z, inp['node_mask'], inp['linker_mask'], bool_mask, inp['context'])
    out_i8 = run_hip_forward(dyn, inp, z, t)
    out_bool = run_hip_forward(dyn, inp, z, t, edge_mask=bool_mask.to(dev()))
    assert rel_l2(ref_bool, ref_i8) > 1e-3
    assert max(report('mask i8', out_i8, ref_i8, z)) <= FWD_TOL and max(report('mask bool', out_bool, ref_bool, z)) <= FWD_TOL


def test_nan_raises_found_nan_exception_with_index_sets():
    from difflinker_amd import FoundNaNException
    nf = 9
    dyn, sd, cfg = make_dynamics(nf, 1, 1, seed=13)
    inp, z, t = ragged_inputs([12, 15, 9], [3, 4, 2], nf, seed=10)
    z = z.clone()
    z[1, 2, 0] = float('nan')          # coordinate NaN in molecule 1 -> radial NaN -> both vel and h
    with pytest.raises(FoundNaNException) as ei:
        run_hip_forward(dyn, inp, z, t)
    with pytest.raises(egnn_oracle.OracleNaN) as eo:
        egnn_oracle.dynamics_forward(sd, cfg, t, z, inp['node_mask'], inp['linker_mask'], inp['edge_mask'], inp['context'])
    assert ei.value.x_h_nan_idx == eo.value.x_h_nan_idx == {1}
    assert ei.value.only_x_nan_idx == eo.value.only_x_nan_idx
    assert ei.value.only_h_nan_idx == eo.value.only_h_nan_idx


@pytest.mark.parametrize('precision', ['f16x3', 'fp32'])
@pytest.mark.parametrize('sizes,linkers', [([116, 10], [5, 2]), ([130, 33, 120], [9, 4, 12])])
def test_molecules_beyond_the_lds_limit_run_on_the_hbm_resident_kernels(sizes, linkers, precision):
    """More than dl_team_max_atoms() = 110 atoms (round 3: 56..110 run fused on a team of compute units, tests/test_gpu_team.py):
    Dynamics.forward switches to dl_egnn_forward_fc_large (the pocket path's per-pass kernels on the dense masked edge
    list, self loops weighted -2 like the int8 mask says) - same numbers."""
    nf, L = 9, 2
    dyn, sd, cfg = make_dynamics(nf, 1, L, seed=14, precision=precision)
    inp, z, t = ragged_inputs(sizes, linkers, nf, seed=11)
    assert not dyn.fits_lds(inp['node_mask'].to(dev()))
    ref = egnn_oracle.dynamics_forward(sd, cfg, t, z, inp['node_mask'], inp['linker_mask'], inp['edge_mask'],
                                       inp['context'])
    out = run_hip_forward(dyn, inp, z, t)
    ev, eh = report(f'large fwd sizes={sizes} {precision}', out, ref, z)
    nm = inp['node_mask'].float()
    assert float((out * (1 - nm)).abs().max()) == 0.0
    assert ev <= FWD_TOLS[precision] and eh <= FWD_TOLS[precision]
    assert torch.equal(out, run_hip_forward(dyn, inp, z, t)), 'bitwise repeatable'
    # the same molecules that DO fit give the same answer on either path (first 10-atom molecule of the small case)
    if sizes == [116, 10]:
        small, zs, ts = ragged_inputs([10, 12], [2, 3], nf, seed=12)
        a = run_hip_forward(dyn, small, zs, ts)
        d = dev()
        b, _ = dyn._launch_forward(ts.to(d), zs.to(d), small['node_mask'].to(d), small['linker_mask'].to(d),
                                   small['edge_mask'].to(d), small['context'].to(d), large=True)
        assert rel_l2(b.cpu(), a) <= 2e-6


@pytest.mark.parametrize('sizes,linkers', [([118, 12], [6, 3]),                      # mixed: one molecule beyond every fused path
                                           ([20, 70, 35, 120, 12, 56], [4, 9, 5, 8, 3, 6]),  # all three size classes, interleaved
                                           ([58, 61], [6, 7]),                      # every molecule needs a team
                                           ([118, 121], [6, 7])])                   # every molecule beyond the fused paths
def test_chain_with_large_molecules_splits_the_batch(sizes, linkers):
    """Molecules of up to dl_max_atoms() = 55 atoms take the fused chain on one compute unit (or a team) each, up to
    dl_team_max_atoms() = 110 the fused chain on a team of at least two, bigger ones the HBM-resident kernels and the
    host-driven loop; noise rows and per-step scalars are those of the whole batch: the reference's numbers."""
    from difflinker_amd import EDM
    nf, T, keep = 8, 6, 2
    dyn, sd, cfg = make_dynamics(nf, 1, 1, seed=33)
    inp, _, _ = ragged_inputs(sizes, linkers, nf, seed=34)
    B, N = inp['x'].shape[:2]
    edm = EDM(dyn, in_node_nf=nf, n_dims=3, timesteps=500, noise_schedule='polynomial_2', noise_precision=1e-5,
              loss_type='l2', norm_values=[1, 4, 10]).to(dev())
    edm.T = T
    bank = edm_oracle.NoiseBank.generate(T, B, N, 3, nf, seed=35)
    orc = edm_oracle.EDMOracle(edm_oracle.make_dynamics_oracle(sd, cfg), in_node_nf=nf, timesteps=500)
    orc.T = T
    want = orc.sample_chain(inp['x'], inp['h'], inp['node_mask'], inp['fragment_mask'], inp['linker_mask'],
                            inp['edge_mask'], inp['context'], bank, keep_frames=keep)
    g = {k: v.to(dev()) for k, v in inp.items()}
    args = (g['x'], g['h'], g['node_mask'], g['fragment_mask'], g['linker_mask'], g['edge_mask'], g['context'])
    got = edm.sample_chain(*args, keep_frames=keep, noise_bank=bank.stacked()).cpu()
    check_chain(f'chain with molecules of {sizes} atoms', got, want, inp)
    # the default noise: the reference's torch.randn call sequence for the WHOLE batch, whichever path a molecule takes
    torch.manual_seed(77)
    a = edm.sample_chain(*args, keep_frames=keep).cpu()
    torch.manual_seed(77)
    drawn = edm.draw_noise_bank(B, N, dev())
    assert torch.equal(a, edm.sample_chain(*args, keep_frames=keep, noise_bank=drawn).cpu())
    # in-kernel / per-step Philox draws keyed by the molecule's index in the whole batch
    from oracle import philox_oracle
    edm.noise_source, edm.noise_seed = 'philox', 5
    got_p = edm.sample_chain(*args, keep_frames=keep).cpu()
    assert edm.noise_seed == 6
    rx, rh = philox_oracle.normal_bank(5, B, N, nf, T + 2)
    draws = []
    for k in range(T + 2):
        draws += [rx[k], rh[k]]
    want_p = orc.sample_chain(inp['x'], inp['h'], inp['node_mask'], inp['fragment_mask'], inp['linker_mask'],
                              inp['edge_mask'], inp['context'], edm_oracle.NoiseBank(draws), keep_frames=keep)
    check_chain(f'chain with molecules of {sizes} atoms, Philox noise', got_p, want_p, inp)


def test_sampler_step_kernel_matches_oracle_arithmetic():
    from difflinker_amd import EDM, _lib
    nf = 8
    dyn, sd, cfg = make_dynamics(nf, 1, 1, seed=15)
    edm = EDM(dyn, in_node_nf=nf, n_dims=3, timesteps=500, noise_schedule='polynomial_2', noise_precision=1e-5,
              loss_type='l2', norm_values=[1, 4, 10])
    g = torch.Generator().manual_seed(2)
    B, N, D = 5, 13, 3 + nf
    z_t, eps, noise = (torch.randn((B, N, D), generator=g) for _ in range(3))
    lm = (torch.rand((B, N, 1), generator=g) > 0.5).float()
    fm = 1 - lm
    coefs, _ = edm.step_coefficients(B)
    t_, a_, c_, s_ = (float(v) for v in coefs[40])
    want = z_t * fm + ((z_t / a_ - c_ * (eps * lm)) + s_ * (noise * lm)) * lm
    d = dev()
    got = edm._sampler_step(z_t.to(d), eps.to(d), noise.to(d), fm.to(d), lm.to(d), _lib.DLStepCoef(t_, a_, c_, s_)).cpu()
    assert max_abs(got, want) <= 2e-6 * float(want.abs().max())


# ---------------------------------------------------------------------------------------------------
def chain_case(nf, n_layers, sizes, linkers, T, keep, seed, timesteps=500, precision=None, coord_gain=0.02, team=None):
    from difflinker_amd import EDM
    dyn, sd, cfg = make_dynamics(nf, 1, n_layers, seed=seed, precision=precision, coord_gain=coord_gain)
    if team is not None:
        dyn.team = team
    inp, _, _ = ragged_inputs(sizes, linkers, nf, seed=seed + 1)
    B, N = inp['x'].shape[:2]
    edm = EDM(dyn, in_node_nf=nf, n_dims=3, timesteps=timesteps, noise_schedule='polynomial_2',
              noise_precision=1e-5, loss_type='l2', norm_values=[1, 4, 10]).to(dev())
    edm.T = T
    bank = edm_oracle.NoiseBank.generate(T, B, N, 3, nf, seed=seed + 2)
    orc = edm_oracle.EDMOracle(edm_oracle.make_dynamics_oracle(sd, cfg), in_node_nf=nf, timesteps=timesteps)
    orc.T = T
    want = orc.sample_chain(inp['x'], inp['h'], inp['node_mask'], inp['fragment_mask'], inp['linker_mask'],
                            inp['edge_mask'], inp['context'], bank, keep_frames=keep)
    g = {k: v.to(dev()) for k, v in inp.items()}
    got = edm.sample_chain(g['x'], g['h'], g['node_mask'], g['fragment_mask'], g['linker_mask'], g['edge_mask'],
                           g['context'], keep_frames=keep, noise_bank=bank.stacked()).cpu()
    return got, want, inp


def check_chain(tag, got, want, inp):
    assert got.shape == want.shape
    lm = inp['linker_mask']
    ex = rel_l2(got[0, :, :, :3] * lm, want[0, :, :, :3] * lm)
    mism = int((got[0, :, :, 3:] != want[0, :, :, 3:]).any(-1).sum())
    efr = rel_l2(got[1:], want[1:]) if got.shape[0] > 1 else 0.0
    print(f'[{tag}] final linker-x rel-L2 {ex:.3e}, one-hot mismatches {mism}, other frames rel-L2 {efr:.3e}')
    assert ex <= CHAIN_TOL and efr <= CHAIN_TOL
    assert mism == 0
    fm = inp['fragment_mask']
    assert max_abs(got[0, :, :, :3] * fm, want[0, :, :, :3] * fm) <= 1e-6     # fragments never move


@pytest.mark.parametrize('precision', ['f16x3', 'fp32'])
def test_chain_vs_oracle_short(precision):
    got, want, inp = chain_case(nf=8, n_layers=2, sizes=[12, 7, 10], linkers=[4, 2, 3], T=12, keep=3, seed=40,
                                precision=precision)
    check_chain(f'chain T=12 {precision}', got, want, inp)


def test_chain_vs_reference_golden(golden_dir):
    from difflinker_amd import EDM
    g = load_golden(golden_dir, 'fc_chain')
    dyn, sd, cfg = make_dynamics(g['nf'], g['ctx'], g['n_layers'], seed=g['weight_seed'], coord_gain=g['coord_gain'])
    edm = EDM(dyn, in_node_nf=g['nf'], n_dims=3, timesteps=500, noise_schedule='polynomial_2',
              noise_precision=1e-5, loss_type='l2', norm_values=[1, 4, 10]).to(dev())
    edm.T = g['T']
    d = dev()
    got = edm.sample_chain(g['x'].to(d), g['h'].to(d), g['node_mask'].to(d), g['fragment_mask'].to(d),
                           g['linker_mask'].to(d), g['edge_mask'].to(d), g['context'].to(d),
                           keep_frames=g['keep_frames'], noise_bank=(g['noise_x'], g['noise_h'])).cpu()
    check_chain('golden fc_chain', got, g['chain'], {'linker_mask': g['linker_mask'], 'fragment_mask': g['fragment_mask']})


def test_chain_keep_frames_all_and_one():
    for keep in (1, 20):
        got, want, inp = chain_case(nf=9, n_layers=1, sizes=[16, 9], linkers=[4, 3], T=20, keep=keep, seed=50)
        check_chain(f'chain T=20 keep={keep}', got, want, inp)


def test_chain_T_differs_from_table_length():
    """--n_steps only overwrites edm.T; the gamma table keeps its trained length (generate.py:103-104)."""
    got, want, inp = chain_case(nf=8, n_layers=1, sizes=[10, 13], linkers=[3, 4], T=7, keep=2, seed=60, timesteps=1000)
    check_chain('chain T=7 on 1000-entry table', got, want, inp)


def test_chain_full_length_zinc_like():
    """C1-like: ZINC hparams (8 blocks), T=50 on the 500-entry table, B=8, N=30."""
    from difflinker_amd import synthetic
    got, want, inp = chain_case(nf=8, n_layers=8, sizes=[30, 24, 27, 29, 25, 30, 26, 28],
                                linkers=[5, 3, 8, 4, 6, 7, 3, 5], T=50, keep=1, seed=70)
    check_chain('chain C1-like T=50 L=8', got, want, inp)


def test_ddpm_sample_chain_end_to_end():
    """DDPM.sample_chain: templates -> context -> COM removal -> fused chain (lightning.py:405-463)."""
    from difflinker_amd import DDPM, synthetic
    hp = dict(in_node_nf=8, n_dims=3, context_node_nf=1, hidden_nf=128, activation='silu', tanh=False, n_layers=2,
              attention=False, norm_constant=1e-6, inv_sublayers=2, sin_embedding=False, normalization_factor=100,
              aggregation_method='sum', diffusion_steps=500, diffusion_noise_schedule='polynomial_2',
              diffusion_noise_precision=1e-5, diffusion_loss_type='l2', normalize_factors=[1, 4, 10],
              include_charges=False, model='egnn_dynamics', data_path='d', train_data_prefix='zinc_final_train',
              val_data_prefix='zinc_final_val', batch_size=8, lr=2e-4, torch_device='cuda:0', test_epochs=20,
              n_stability_samples=10, normalization='batch_norm', anchors_context=False)
    torch.manual_seed(0)
    m = DDPM(**hp).to(dev()).eval()
    m.edm.T = 6
    data, cfg = synthetic.make_batch('C1', seed=2, batch=4, device=dev())
    torch.manual_seed(123)
    chain, node_mask = m.sample_chain(data, keep_frames=1)
    torch.manual_seed(123)
    chain2, _ = m.sample_chain(data, keep_frames=1)
    assert chain.shape == (1, 4, data['positions'].shape[1], 11)
    assert torch.equal(chain, chain2), 'same torch seed -> same sample'
    x, h = chain[0][..., :3], chain[0][..., 3:]
    nm = node_mask.float()
    assert torch.isfinite(chain).all()
    assert float((chain[0] * (1 - nm)).abs().max()) == 0.0
    assert torch.equal(h.sum(-1), nm.squeeze(-1)), 'one-hot rows for real atoms, zero rows for padding'
    # fragments keep their (centred) input coordinates
    from difflinker_amd import utils
    x_in = utils.remove_partial_mean_with_mask(data['positions'] * data['fragment_mask'], nm, data['fragment_mask'])
    fm = data['fragment_mask']
    assert max_abs((x * fm).cpu(), (x_in * fm).cpu()) <= 1e-5


def test_chain_full_length_geom_like():
    """GEOM hparams (6 blocks), the full T=500 chain on a few C2-sized molecules against the oracle."""
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    # coordinate head at the reference's own init scale (xavier gain 0.001, egnn.py:90-91): the 0.02 'lively'
    # variant used elsewhere overflows to NaN within 500 steps on the oracle itself
    got, want, inp = chain_case(nf=9, n_layers=6, sizes=[50, 41], linkers=[8, 6], T=500, keep=1, seed=90, coord_gain=0.001)
    check_chain('chain GEOM-like T=500 L=6', got, want, inp)


def test_geom_sized_forward_full_batch():
    """BASELINE config C2 at full size (B=256, N=50, 6 blocks): one forward against the oracle."""
    from difflinker_amd import synthetic
    nf, L = 9, 6
    dyn, sd, cfg = make_dynamics(nf, 1, L, seed=80)
    data, c2 = synthetic.make_batch('C2', seed=1)
    inp = synthetic.sampler_inputs(data)
    B, N = inp['x'].shape[:2]
    g = torch.Generator().manual_seed(4)
    z = torch.cat([inp['x'], inp['h']], dim=2) * inp['fragment_mask'] + \
        torch.randn((B, N, 3 + nf), generator=g) * inp['linker_mask']
    t = torch.full((B, 1), 0.37)
    torch.set_num_threads(min(16, os.cpu_count() or 1))      # more threads only slow the CPU oracle down
    ref = egnn_oracle.dynamics_forward(sd, cfg, t, z, inp['node_mask'], inp['linker_mask'], inp['edge_mask'], inp['context'])
    out = run_hip_forward(dyn, inp, z, t)
    ev, eh = report('C2 full forward', out, ref, z)
    assert ev <= FWD_TOL and eh <= FWD_TOL


def test_geom_sized_chain_full_batch_properties():
    """BASELINE config C2 at full size (B=256, N=50, 6 blocks, T=500) through size-independent properties: the chain
    is bitwise repeatable, a molecule's sample does not depend on the rest of the batch (a sub-batch with the same noise
    rows gives the same rows), fragments never move, every real atom ends as a one-hot row, padding stays zero, and the
    linker moves rigidly with a rotation + translation of the input (E(3) equivariance of the whole sampler)."""
    from difflinker_amd import EDM, synthetic
    nf, L, T = 9, 6, 500
    dyn, sd, cfg = make_dynamics(nf, 1, L, seed=81, coord_gain=0.001)
    data, _ = synthetic.make_batch('C2', seed=2)
    inp = {k: v.to(dev()) for k, v in synthetic.sampler_inputs(data).items()}
    B, N = inp['x'].shape[:2]
    edm = EDM(dyn, in_node_nf=nf, n_dims=3, timesteps=500, noise_schedule='polynomial_2', noise_precision=1e-5,
              loss_type='l2', norm_values=[1, 4, 10]).to(dev())
    edm.noise_source = 'philox'

    def run(sel=slice(None), x=None, seed=3, mol_offset=0):
        edm.noise_seed = seed
        em = inp['edge_mask'].view(B, N * N)[sel].reshape(-1, 1)
        return edm.sample_chain((inp['x'] if x is None else x)[sel], inp['h'][sel], inp['node_mask'][sel],
                                inp['fragment_mask'][sel], inp['linker_mask'][sel], em, inp['context'][sel],
                                keep_frames=1, mol_offset=mol_offset)[0]
    # (a) one launch per chain (split_chain off): a molecule's sample is bit for bit independent of its batch-mates
    edm.split_chain = False
    a1 = run()
    edm.coef_batch = edm.team_batch = B     # per-step scalars and team size of the whole batch, as a shard would
    sub = run(slice(100, 108), mol_offset=100)
    edm.coef_batch = edm.team_batch = None
    assert torch.equal(sub, a1[100:108]), 'independent of the rest of the batch'
    # (b) the default since round 5: the chain of a ragged batch that fills the chip runs in two launches (EDM.split_chain: the
    # big molecules finish on teams of two in the compute units the small ones left); the calls on a team sum messages in the
    # team's order, so the two agree to fp32 rounding - and the plan, a function of the sizes, is repeatable bit for bit
    edm.split_chain = True
    a = run()
    assert torch.isfinite(a).all()
    assert torch.equal(a, run()), 'bitwise repeatable'
    lm_ = inp['linker_mask']
    drift = rel_l2(a[..., :3] * lm_, a1[..., :3] * lm_)
    print(f'[C2 full batch, T=500] two launches (split chain) vs one: final linker-x rel-L2 {drift:.3e}, '
          f'atom-type mismatches {int((a[..., 3:] != a1[..., 3:]).any(-1).sum())}')
    assert drift <= 1e-5 and torch.equal(a[..., 3:], a1[..., 3:])
    nm, fm, lm = inp['node_mask'].float(), inp['fragment_mask'], inp['linker_mask']
    assert float((a * (1 - nm)).abs().max()) == 0.0
    assert torch.equal(a[..., 3:].sum(-1), nm.squeeze(-1))
    assert float(((a[..., :3] - inp['x']) * fm).abs().max()) <= 1e-5
    # rigid motion of the input (fragments stay centred: rotate about the origin; the sampler works in that frame)
    g = torch.Generator().manual_seed(9)
    q, _ = torch.linalg.qr(torch.randn(3, 3, generator=g))
    q = (q * torch.sign(torch.linalg.det(q))).to(dev())
    b = run(x=inp['x'] @ q.T)
    # noise is drawn in the lab frame, so only the fragment part is equivariant sample by sample; the linker must stay
    # a valid, finite sample attached to the rotated fragments
    assert float(((b[..., :3] - inp['x'] @ q.T) * fm).abs().max()) <= 1e-5
    assert torch.isfinite(b).all() and torch.equal(b[..., 3:].sum(-1), nm.squeeze(-1))
    assert float(((b[..., :3] * lm).norm(dim=-1) - (a[..., :3] * lm).norm(dim=-1)).abs().mean()) < 5.0


# ---------------------------------------------------------------------------------------------------
# pocket-conditioned path (DynamicsWithPockets, radius graph)
def make_pocket_dynamics(nf, n_layers, seed, graph_type='FC-10A-4A', coord_gain=0.02, precision=None):
    from difflinker_amd import DynamicsWithPockets
    dyn = DynamicsWithPockets(n_dims=3, in_node_nf=nf, context_node_nf=2, hidden_nf=128, n_layers=n_layers,
                              norm_constant=1e-6, normalization='batch_norm', graph_type=graph_type)
    sd = seeded_state_dict(nf + 3, 128, n_layers, seed, coord_gain=coord_gain)
    dyn.load_state_dict(sd, strict=True)
    if precision is not None:
        dyn.precision = precision
    cfg = EGNNConfig(in_node_nf=nf, context_node_nf=2, n_layers=n_layers, graph_type=graph_type)
    return dyn.to(dev()), sd, cfg


def pocket_inputs(batch, n_frag, n_pocket, linker, nf, seed):
    from difflinker_amd import synthetic
    from difflinker_amd.datasets import collate
    data = collate(synthetic.pocket_molecules(batch, n_frag, n_pocket, linker, nf, seed))
    inp = synthetic.sampler_inputs(data, pockets=True)
    B, N = inp['x'].shape[:2]
    g = torch.Generator().manual_seed(seed + 1)
    z = torch.cat([inp['x'], inp['h']], dim=2) * inp['fragment_mask'] + \
        torch.cat([2.0 * torch.randn((B, N, 3), generator=g), torch.randn((B, N, nf), generator=g)], dim=2) * inp['linker_mask']
    t = torch.rand((B, 1), generator=g)
    return inp, z, t


@pytest.mark.parametrize('precision', ['f16x3', 'fp32'])
def test_pocket_forward_vs_reference_golden(golden_dir, precision):
    g = load_golden(golden_dir, 'pocket_forward')
    dyn, sd, cfg = make_pocket_dynamics(g['nf'], g['n_layers'], seed=g['weight_seed'], coord_gain=g['coord_gain'],
                                        precision=precision)
    inp = {k: g[k] for k in ('node_mask', 'linker_mask', 'edge_mask', 'context')}
    out = run_hip_forward(dyn, inp, g['xh'], g['t'])
    ev, eh = report(f'golden pocket_forward {precision}', out, g['out'], g['xh'])
    assert ev <= FWD_TOLS[precision] and eh <= FWD_TOLS[precision]
    nm = g['node_mask'].float()
    assert float((out * (1 - nm)).abs().max()) == 0.0


@pytest.mark.parametrize('precision', ['f16x3', 'fp32'])
@pytest.mark.parametrize('graph_type', ['FC-10A-4A', 'FC-4A', '4A'])
def test_pocket_forward_vs_oracle(graph_type, precision):
    nf = 9
    dyn, sd, cfg = make_pocket_dynamics(nf, 3, seed=31, graph_type=graph_type, precision=precision)
    inp, z, t = pocket_inputs(batch=3, n_frag=12, n_pocket=70, linker=(4, 9), nf=nf, seed=33)
    ref = egnn_oracle.dynamics_forward_pockets(sd, cfg, t, z, inp['node_mask'], inp['linker_mask'], inp['edge_mask'],
                                               inp['context'])
    out = run_hip_forward(dyn, inp, z, t)
    ev, eh = report(f'pocket fwd {graph_type} {precision}', out, ref, z)
    assert ev <= FWD_TOLS[precision] and eh <= FWD_TOLS[precision]
    a = run_hip_forward(dyn, inp, z, t)
    assert torch.equal(a, out), 'pocket path must be bitwise repeatable'


def test_pocket_chain_vs_oracle():
    """EDM.sample_chain on the pocket path (host-driven loop: HIP denoiser + fused HIP tail per step)."""
    from difflinker_amd import EDM
    nf, T = 9, 6
    dyn, sd, cfg = make_pocket_dynamics(nf, 2, seed=35)
    inp, _, _ = pocket_inputs(batch=2, n_frag=10, n_pocket=40, linker=(3, 6), nf=nf, seed=37)
    B, N = inp['x'].shape[:2]
    edm = EDM(dyn, in_node_nf=nf, n_dims=3, timesteps=500, noise_schedule='polynomial_2', noise_precision=1e-5,
              loss_type='l2', norm_values=[1, 4, 10]).to(dev())
    edm.T = T
    bank = edm_oracle.NoiseBank.generate(T, B, N, 3, nf, seed=39)
    orc = edm_oracle.EDMOracle(edm_oracle.make_dynamics_oracle(sd, cfg), in_node_nf=nf, timesteps=500)
    orc.T = T
    want = orc.sample_chain(inp['x'], inp['h'], inp['node_mask'], inp['fragment_mask'], inp['linker_mask'],
                            inp['edge_mask'], inp['context'], bank, keep_frames=2)
    g = {k: v.to(dev()) for k, v in inp.items()}
    got = edm.sample_chain(g['x'], g['h'], g['node_mask'], g['fragment_mask'], g['linker_mask'], g['edge_mask'],
                           g['context'], keep_frames=2, noise_bank=bank.stacked()).cpu()
    check_chain('pocket chain T=6', got, want, inp)


# ---------------------------------------------------------------------------------------------------
# InpaintingEDM (edm.py:549-727): centred dynamics, fragments re-drawn from q, centre of gravity projected out
def test_inpainting_chain_vs_reference_golden_and_oracle(golden_dir):
    from difflinker_amd import InpaintingEDM, Dynamics
    g = load_golden(golden_dir, 'inpainting_chain')
    nf, ctx, L, T, keep = g['nf'], g['ctx'], g['n_layers'], g['T'], g['keep_frames']
    dyn = Dynamics(n_dims=3, in_node_nf=nf, context_node_nf=ctx, hidden_nf=128, n_layers=L, norm_constant=1e-6,
                   normalization='batch_norm', centering=True)
    dyn.load_state_dict(seeded_state_dict(nf + ctx + 1, 128, L, g['weight_seed'], coord_gain=g['coord_gain']), strict=True)
    edm = InpaintingEDM(dyn.to(dev()), in_node_nf=nf, n_dims=3, timesteps=500, noise_schedule='polynomial_2',
                        noise_precision=1e-5, loss_type='l2', norm_values=[1, 4, 10]).to(dev())
    edm.T = T
    d = dev()
    got = edm.sample_chain(g['x'].to(d), g['h'].to(d), g['node_mask'].to(d), g['edge_mask'].to(d),
                           g['fragment_mask'].to(d), g['linker_mask'].to(d), g['context'].to(d), keep_frames=keep,
                           noise_bank=(g['noise_x'], g['noise_h'])).cpu()
    want = g['chain']
    nm = g['node_mask'].float()
    ex = rel_l2(got[..., :3], want[..., :3])
    mism = int((got[0, :, :, 3:] != want[0, :, :, 3:]).any(-1).sum())
    print(f'[inpainting chain T={T}] x rel-L2 {ex:.3e}, one-hot mismatches {mism}, frames rel-L2 {rel_l2(got[1:], want[1:]):.3e}')
    assert got.shape == want.shape and ex <= CHAIN_TOL and rel_l2(got[1:], want[1:]) <= CHAIN_TOL and mism == 0
    assert float((got * (1 - nm)).abs().max()) == 0.0
    assert float((got[1:, :, :, :3] * nm).sum(2).abs().max()) <= 1e-4, 'centre of gravity projected out every step'
    # DDPM(inpainting=True) builds this sampler on a centred denoiser (lightning.py:99-102)
    from difflinker_amd import DDPM
    hp = dict(in_node_nf=8, n_dims=3, context_node_nf=1, hidden_nf=128, activation='silu', tanh=False, n_layers=1,
              attention=False, norm_constant=1e-6, inv_sublayers=2, sin_embedding=False, normalization_factor=100,
              aggregation_method='sum', diffusion_steps=500, diffusion_noise_schedule='polynomial_2',
              diffusion_noise_precision=1e-5, diffusion_loss_type='l2', normalize_factors=[1, 4, 10],
              include_charges=False, model='egnn_dynamics', data_path='d', train_data_prefix='zinc_final_train',
              val_data_prefix='zinc_final_val', batch_size=8, lr=2e-4, torch_device='cuda:0', test_epochs=20,
              n_stability_samples=10, normalization='batch_norm', anchors_context=False, inpainting=True)
    m = DDPM(**hp).to(d).eval()
    assert isinstance(m.edm, InpaintingEDM) and m.edm.dynamics.centering
    m.edm.T = 4
    from difflinker_amd import synthetic
    data, _ = synthetic.make_batch('C1', seed=2, batch=3, device=d)
    torch.manual_seed(5)
    chain, node_mask = m.sample_chain(data, keep_frames=1)
    assert chain.shape[1:3] == data['positions'].shape[:2] and torch.isfinite(chain).all()
    assert torch.equal(chain[0][..., 3:].sum(-1), node_mask.squeeze(-1).float())


@pytest.mark.parametrize('team', [1, 'auto'])
@pytest.mark.parametrize('precision', ['f16x3', 'fp32'])
def test_coordinate_pass_evaluates_the_linker_mask_receivers_only(team, precision):
    """The kernels sum the coordinate head only for receiving atoms with linker_mask != 0 (the reference multiplies the
    other sums by zero, egnn.py:113-116).  Edge cases of that list: no linker atom at all (nothing moves: velocity exactly
    zero), every atom a linker atom, one linker atom, a fractional mask (a weight, not a switch), beside ordinary molecules."""
    nf, L = 9, 2
    dyn, sd, cfg = make_dynamics(nf, 1, L, seed=211, precision=precision)
    dyn.team = team
    sizes, linkers = [30, 22, 41, 17, 55, 9], [5, 21, 1, 3, 12, 2]
    inp, z, t = ragged_inputs(sizes, linkers, nf, seed=212)
    lm = inp['linker_mask'].clone()
    lm[0] = 0.0                                            # molecule 0: no linker atom
    lm[1, :sizes[1]] = 1.0                                 # molecule 1: every atom moves
    lm[3, :sizes[3]] *= 0.5                                # molecule 3: its linker atoms weigh one half
    lm[3, 0] = 0.25                                        # ... and a fragment atom moves a little too
    inp = dict(inp, linker_mask=lm)
    ref = egnn_oracle.dynamics_forward(sd, cfg, t, z, inp['node_mask'], lm, inp['edge_mask'], inp['context'])
    out = run_hip_forward(dyn, inp, z, t)
    ev, eh = report(f'receiver list edge cases, team {team} {precision}', out, ref, z)
    assert ev <= FWD_TOLS[precision] and eh <= FWD_TOLS[precision]
    assert float(out[0, :, :3].abs().max()) == 0.0, 'no linker atom: nothing moves'
    moved = out[..., :3].abs().sum(-1) > 0
    assert not bool((moved & (lm.squeeze(-1) == 0)).any()), 'atoms outside the mask never move'
    assert bool(moved[1, :sizes[1]].all()) and bool(moved[3, 0]) and int(moved[2].sum()) == 1
    assert torch.equal(out, run_hip_forward(dyn, inp, z, t))
