"""GPU tests of the counter-based in-kernel noise: the device stream equals its CPU restatement
(oracle/philox_oracle.py, itself checked against the Random123 known-answer vectors), a chain drawn in the kernel
matches the CPU oracle fed with that stream, and samples do not depend on how the batch is split."""
import pytest
import torch

from helpers import rel_l2, max_abs
from oracle import edm_oracle, philox_oracle
from test_gpu_parity import make_dynamics, ragged_inputs, check_chain, dev

pytestmark = pytest.mark.gpu


def make_edm(nf, n_layers, T, seed, timesteps=500):
    from difflinker_amd import EDM
    dyn, sd, cfg = make_dynamics(nf, 1, n_layers, seed=seed)
    edm = EDM(dyn, in_node_nf=nf, n_dims=3, timesteps=timesteps, noise_schedule='polynomial_2', noise_precision=1e-5,
              loss_type='l2', norm_values=[1, 4, 10]).to(dev())
    edm.T = T
    return edm, sd, cfg


def test_device_bank_equals_cpu_restatement():
    edm, _, _ = make_edm(9, 1, T=6, seed=1)
    nx, nh = edm.philox_noise_bank(5, 23, dev(), mol_offset=7, seed=0x1234567890ABCDEF)
    rx, rh = philox_oracle.normal_bank(0x1234567890ABCDEF, 5, 23, 9, 8, mol_offset=7)
    ex, eh = max_abs(nx.cpu(), rx), max_abs(nh.cpu(), rh)
    print(f'[philox bank] max-abs x {ex:.2e} h {eh:.2e}')
    # same integers, fp32 log / sqrt / sincos on both sides: a few ulp of |value| <= 5.5
    assert ex <= 4e-6 and eh <= 4e-6
    assert edm.noise_seed == 0, 'an explicit seed does not advance the generator'


@pytest.mark.parametrize('precision', ['f16x3', 'fp32'])
def test_chain_with_in_kernel_noise_matches_oracle(precision):
    nf, T, keep = 8, 12, 3
    edm, sd, cfg = make_edm(nf, 2, T, seed=40)
    edm.dynamics.precision = precision
    inp, _, _ = ragged_inputs([12, 7, 10], [4, 2, 3], nf, seed=41)
    B, N = inp['x'].shape[:2]
    edm.noise_source, edm.noise_seed = 'philox', 99
    g = {k: v.to(dev()) for k, v in inp.items()}
    got = edm.sample_chain(g['x'], g['h'], g['node_mask'], g['fragment_mask'], g['linker_mask'], g['edge_mask'],
                           g['context'], keep_frames=keep).cpu()
    assert edm.noise_seed == 100, 'one chain consumes one seed'
    rx, rh = philox_oracle.normal_bank(99, B, N, nf, T + 2)
    draws = []
    for k in range(T + 2):
        draws += [rx[k], rh[k]]
    orc = edm_oracle.EDMOracle(edm_oracle.make_dynamics_oracle(sd, cfg), in_node_nf=nf, timesteps=500)
    orc.T = T
    want = orc.sample_chain(inp['x'], inp['h'], inp['node_mask'], inp['fragment_mask'], inp['linker_mask'],
                            inp['edge_mask'], inp['context'], edm_oracle.NoiseBank(draws), keep_frames=keep)
    check_chain(f'chain T=12 in-kernel noise {precision}', got, want, inp)
    # the bank entry point replays the very same stream (what the pocket host loop uses)
    edm.noise_seed = 99
    bank = edm.philox_noise_bank(B, N, dev())
    edm.noise_source = 'torch'
    got2 = edm.sample_chain(g['x'], g['h'], g['node_mask'], g['fragment_mask'], g['linker_mask'], g['edge_mask'],
                            g['context'], keep_frames=keep, noise_bank=bank).cpu()
    assert torch.equal(got, got2), 'in-kernel draws == dl_philox_fill bank, bitwise'


def test_samples_do_not_depend_on_the_batch_split():
    nf, T = 9, 8
    edm, _, _ = make_edm(nf, 1, T, seed=3)
    inp, _, _ = ragged_inputs([14, 9, 12, 5], [4, 3, 5, 2], nf, seed=8)
    g = {k: v.to(dev()) for k, v in inp.items()}
    edm.noise_source = 'philox'
    args = lambda lo, hi: dict(x=g['x'][lo:hi], h=g['h'][lo:hi], node_mask=g['node_mask'][lo:hi],       # noqa: E731
                               fragment_mask=g['fragment_mask'][lo:hi], linker_mask=g['linker_mask'][lo:hi],
                               edge_mask=g['edge_mask'].view(4, -1)[lo:hi].reshape(-1, 1), context=g['context'][lo:hi])
    edm.noise_seed = 7
    full = edm.sample_chain(keep_frames=2, **args(0, 4))
    edm.noise_seed = 7
    lo_half = edm.sample_chain(keep_frames=2, **args(0, 2))
    edm.noise_seed = 7
    hi_half = edm.sample_chain(keep_frames=2, mol_offset=2, **args(2, 4))
    assert torch.equal(full[:, :2], lo_half) and torch.equal(full[:, 2:], hi_half)
    edm.noise_seed = 8
    other = edm.sample_chain(keep_frames=2, **args(0, 4))
    assert not torch.equal(full, other)


def test_default_noise_bank_is_the_reference_call_sequence():
    """The default ('torch') bank: 2(T+2) torch.randn calls in the reference's order and shapes
    (utils.sample_gaussian_with_mask via edm.py:136,205,228 -> :328-345), drawn in place."""
    edm, _, _ = make_edm(9, 1, T=5, seed=2)
    B, N = 3, 17
    torch.manual_seed(321)
    nx, nh = edm.draw_noise_bank(B, N, dev())
    torch.manual_seed(321)
    for k in range(5 + 2):
        assert torch.equal(nx[k], torch.randn((B, N, 3), device=dev()))
        assert torch.equal(nh[k], torch.randn((B, N, 9), device=dev()))
