"""Round-5 additions (VERDICT / ADVICE round 4): weights with the statistics of a TRAINED checkpoint through every kernel
family, the molecules beyond one per compute unit on teams in a second launch, the team-failure path of the host-driven
chain, world = 1 against a simulated shard on a mixed-size batch, NaN index sets of the earliest denoiser call only."""
import os

import pytest
import torch

import test_gpu_parity as P
from helpers import rel_l2, seeded_state_dict, trained_like_state_dict
from oracle import edm_oracle, egnn_oracle
from oracle.egnn_oracle import EGNNConfig

pytestmark = pytest.mark.gpu


def _edm(dyn, nf, T, timesteps=500):
    from difflinker_amd import EDM
    edm = EDM(dyn, in_node_nf=nf, n_dims=3, timesteps=timesteps, noise_schedule='polynomial_2', noise_precision=1e-5,
              loss_type='l2', norm_values=[1, 4, 10]).to(P.dev())
    edm.T = T
    return edm


# ---- trained-like weights -----------------------------------------------------------------------------------------------
def test_trained_like_weights_fc_against_fp32_and_fp64_oracles():
    """Every parity case so far used nn.Linear-style random inits, where every row of a matrix has the same norm and the
    a-priori power-of-two scales of f16x3 (row-L1 norms x measured maxima) are tight.  Here: log-normal row factors
    (sigma = 1.5), one row x 2^10, biases x 30 (helpers.trained_like_state_dict: row-L1 norms over 17 binades, bias maxima in
    the thousands) on a C2-shaped batch, both arithmetic modes against the fp32 oracle AND the fp64 oracle: the split scheme
    must stay within twice the exact-fp32 mode's own rounding error, i.e. a loose bound must not eat its low bits."""
    from difflinker_amd import synthetic
    nf, L = 9, 6
    data, _ = synthetic.make_batch('C2', seed=1, batch=32)
    inp = synthetic.sampler_inputs(data)
    B, N = inp['x'].shape[:2]
    g = torch.Generator().manual_seed(4)
    z = torch.cat([inp['x'], inp['h']], dim=2) * inp['fragment_mask'] + torch.randn((B, N, 3 + nf), generator=g) * inp['linker_mask']
    t = torch.full((B, 1), 0.37)
    sd = trained_like_state_dict(seeded_state_dict(nf + 2, 128, L, 80, coord_gain=0.02), seed=7)
    cfg = EGNNConfig(in_node_nf=nf, context_node_nf=1, n_layers=L)
    ref32 = egnn_oracle.dynamics_forward(sd, cfg, t, z, inp['node_mask'], inp['linker_mask'], inp['edge_mask'], inp['context'])
    sd64 = {k: v.double() for k, v in sd.items()}
    ref64 = egnn_oracle.dynamics_forward(sd64, cfg, t.double(), z.double(), inp['node_mask'], inp['linker_mask'].double(),
                                         inp['edge_mask'], inp['context'].double())
    err = {}
    for precision in ('fp32', 'f16x3'):
        dyn, _, _ = P.make_dynamics(nf, 1, L, seed=80, precision=precision)
        dyn.load_state_dict(sd, strict=True)
        dyn.invalidate_packed()
        out = P.run_hip_forward(dyn, inp, z, t).double()
        err[precision] = (rel_l2(out[..., 3:], ref64[..., 3:]), float((out[..., :3] - ref64[..., :3]).norm()),
                          rel_l2(out[..., 3:], ref32[..., 3:].double()), rel_l2(out[..., :3], ref32[..., :3].double()))
    o32 = (rel_l2(ref32[..., 3:].double(), ref64[..., 3:]), float((ref32[..., :3].double() - ref64[..., :3]).norm()))
    print(f'[trained-like FC vs fp64] h rel-L2: fp32 oracle {o32[0]:.3e}, fp32 mode {err["fp32"][0]:.3e}, f16x3 {err["f16x3"][0]:.3e}; '
          f'vel abs-L2: fp32 oracle {o32[1]:.3e}, fp32 mode {err["fp32"][1]:.3e}, f16x3 {err["f16x3"][1]:.3e}; '
          f'vs the fp32 oracle: h {err["f16x3"][2]:.3e} / raw vel {err["f16x3"][3]:.3e} (f16x3), h {err["fp32"][2]:.3e} / {err["fp32"][3]:.3e} (fp32 mode)')
    assert err['f16x3'][0] <= 2.0 * err['fp32'][0]
    assert err['f16x3'][1] <= 2.0 * err['fp32'][1]
    for precision in ('fp32', 'f16x3'):
        assert err[precision][2] <= P.FWD_TOLS[precision] and err[precision][3] <= 1e-4


def test_trained_like_weights_pocket_kernels():
    """The same weights through the radius-graph kernels (egnn_sparse.hip) at the C4 geometry."""
    nf, L = 9, 6
    dyn, sd0, cfg = P.make_pocket_dynamics(nf, L, seed=131)
    sd = trained_like_state_dict(sd0, seed=9)
    inp, z, t = P.pocket_inputs(batch=2, n_frag=30, n_pocket=250, linker=(6, 12), nf=nf, seed=133)
    ref32 = egnn_oracle.dynamics_forward_pockets(sd, cfg, t, z, inp['node_mask'], inp['linker_mask'], inp['edge_mask'], inp['context'])
    sd64 = {k: v.double() for k, v in sd.items()}
    ref64 = egnn_oracle.dynamics_forward_pockets(sd64, cfg, t.double(), z.double(), inp['node_mask'], inp['linker_mask'].double(),
                                                 inp['edge_mask'], inp['context'].double())
    err = {}
    for precision in ('fp32', 'f16x3'):
        dyn.precision = precision
        dyn.load_state_dict(sd, strict=True)
        dyn.invalidate_packed()
        out = P.run_hip_forward(dyn, inp, z, t).double()
        err[precision] = (rel_l2(out[..., 3:], ref64[..., 3:]), float((out[..., :3] - ref64[..., :3]).norm()),
                          rel_l2(out[..., 3:], ref32[..., 3:].double()))
    print(f'[trained-like pockets vs fp64] h rel-L2: fp32 mode {err["fp32"][0]:.3e}, f16x3 {err["f16x3"][0]:.3e}; vel abs-L2: '
          f'fp32 mode {err["fp32"][1]:.3e}, f16x3 {err["f16x3"][1]:.3e}; h vs the fp32 oracle: {err["f16x3"][2]:.3e} (f16x3)')
    assert err['f16x3'][0] <= 2.0 * err['fp32'][0]
    assert err['f16x3'][1] <= 2.0 * err['fp32'][1]
    assert err['f16x3'][2] <= P.FWD_TOLS['f16x3'] and err['fp32'][2] <= P.FWD_TOLS['fp32']


def test_trained_like_weights_chain():
    """...and along a chain: T = 30 with a shared noise bank against the oracle (the sampler feeds each step's rounding into
    the next forward's bounds)."""
    from difflinker_amd import EDM
    nf, L, T = 8, 2, 30
    dyn, sd0, cfg = P.make_dynamics(nf, 1, L, seed=61)
    sd = trained_like_state_dict(sd0, seed=11)
    dyn.load_state_dict(sd, strict=True)
    dyn.invalidate_packed()
    inp, _, _ = P.ragged_inputs([40, 33, 50, 12], [7, 5, 9, 3], nf, seed=62)
    B, N = inp['x'].shape[:2]
    edm = _edm(dyn, nf, T)
    bank = edm_oracle.NoiseBank.generate(T, B, N, 3, nf, seed=63)
    orc = edm_oracle.EDMOracle(edm_oracle.make_dynamics_oracle(sd, cfg), in_node_nf=nf, timesteps=500)
    orc.T = T
    want = orc.sample_chain(inp['x'], inp['h'], inp['node_mask'], inp['fragment_mask'], inp['linker_mask'], inp['edge_mask'],
                            inp['context'], bank, keep_frames=3)
    assert torch.isfinite(want).all()
    g = {k: v.to(P.dev()) for k, v in inp.items()}
    got = edm.sample_chain(g['x'], g['h'], g['node_mask'], g['fragment_mask'], g['linker_mask'], g['edge_mask'], g['context'],
                           keep_frames=3, noise_bank=bank.stacked()).cpu()
    P.check_chain('trained-like weights, chain T=30', got, want, inp)


# ---- molecules beyond one per compute unit ---------------------------------------------------------------------------------
def test_molecules_beyond_one_per_compute_unit_are_sampled_by_teams_beside_the_rest():
    """A batch of (compute units + 3) molecules: ``EDM.sample_chain`` puts the three smallest on teams of four in a second launch
    on another stream (dl_chain_args.order_first / order_count) instead of letting them wait for a free compute unit.  The chain
    must be the oracle's; the molecules of the first launch are bit for bit what the one-launch path samples (the switch
    ``overflow_teams = False``), the three on teams agree to fp32 rounding; repeatable bit for bit."""
    nf, L, T = 8, 1, 5
    cus = torch.cuda.get_device_properties(P.dev()).multi_processor_count
    B = cus + 3
    g0 = torch.Generator().manual_seed(5)
    sizes = torch.randint(4, 13, (B,), generator=g0).tolist()
    linkers = [max(1, s // 4) for s in sizes]
    dyn, sd, cfg = P.make_dynamics(nf, 1, L, seed=71)
    dyn.team = 'auto'
    inp, _, _ = P.ragged_inputs(sizes, linkers, nf, seed=72)
    N = inp['x'].shape[1]
    edm = _edm(dyn, nf, T)
    bank = edm_oracle.NoiseBank.generate(T, B, N, 3, nf, seed=73)
    orc = edm_oracle.EDMOracle(edm_oracle.make_dynamics_oracle(sd, cfg), in_node_nf=nf, timesteps=500)
    orc.T = T
    want = orc.sample_chain(inp['x'], inp['h'], inp['node_mask'], inp['fragment_mask'], inp['linker_mask'], inp['edge_mask'],
                            inp['context'], bank, keep_frames=2)
    g = {k: v.to(P.dev()) for k, v in inp.items()}

    def run():
        out = edm.sample_chain(g['x'], g['h'], g['node_mask'], g['fragment_mask'], g['linker_mask'], g['edge_mask'], g['context'],
                               keep_frames=2, noise_bank=bank.stacked())
        torch.cuda.synchronize()
        return out.cpu()
    assert edm.overflow_teams
    got = run()
    P.check_chain(f'B = {B} on {cus} compute units, 3 molecules on teams beside the rest', got, want, inp)
    assert torch.equal(got, run()), 'bitwise repeatable'
    edm.overflow_teams = False
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('ignore', RuntimeWarning)
        plain = run()
    order = torch.argsort(torch.tensor(sizes), descending=True, stable=True)
    first, over = order[:cus], order[cus:]
    assert torch.equal(got[:, first], plain[:, first]), 'molecules of the first launch: the same bits as without the second launch'
    lm = inp['linker_mask'][over]
    print(f'molecules on teams vs one compute unit each: linker-x rel-L2 {rel_l2(got[0, over, :, :3] * lm, plain[0, over, :, :3] * lm):.3e}')
    assert rel_l2(got[0, over, :, :3] * lm, plain[0, over, :, :3] * lm) <= 1e-5
    assert torch.equal(got[0, over, :, 3:], plain[0, over, :, 3:])


# ---- host-driven chain: a team that does not assemble ----------------------------------------------------------------------
def test_host_loop_chain_survives_a_team_failure_with_the_same_draws():
    """a ``centering=True`` denoiser runs ``EDM.sample_chain`` on the host-driven loop, where molecules of 56..110
    atoms take teams.  The first team launch is made to fail (test-hooks build of the library): ``TeamNotAssembled`` is raised
    after the FIRST denoiser call, the chain is repeated without teams from the restored generator state - with
    ``noise_source = 'torch'`` and no bank - and equals, bit for bit, the chain a denoiser that never uses teams samples from the
    same seed."""
    from difflinker_amd import Dynamics, _lib
    nf, L, T = 8, 1, 4
    dyn = Dynamics(n_dims=3, in_node_nf=nf, context_node_nf=1, hidden_nf=128, n_layers=L, norm_constant=1e-6, centering=True)
    sd = seeded_state_dict(nf + 2, 128, L, 81)
    dyn.load_state_dict(sd, strict=True)
    dyn = dyn.to(P.dev())
    inp, _, _ = P.ragged_inputs([60, 20, 75], [6, 4, 8], nf, seed=82)
    g = {k: v.to(P.dev()) for k, v in inp.items()}
    edm = _edm(dyn, nf, T)
    assert edm.noise_source == 'torch'

    def run():
        torch.manual_seed(99)
        out = edm.sample_chain(g['x'], g['h'], g['node_mask'], g['fragment_mask'], g['linker_mask'], g['edge_mask'], g['context'],
                               keep_frames=1)
        torch.cuda.synchronize()
        return out.cpu()
    dyn._no_teams = True
    want = run()
    dyn._no_teams = False
    with _lib.test_hooks() as lib:
        dyn.invalidate_packed()
        teams = run()                                  # teams assemble
        lib.dl_debug_team_fault(1)                     # the next team launch fails: the first denoiser call of the chain
        got = run()
    dyn.invalidate_packed()
    assert torch.isfinite(got).all()
    assert torch.equal(got, want), 'the repeated chain must draw the same noise and use no teams'
    lm = inp['linker_mask']
    assert rel_l2(teams[0, :, :, :3] * lm, want[0, :, :, :3] * lm) <= 1e-5


# ---- world = 1 against a shard --------------------------------------------------------------------------------------------
@pytest.mark.parametrize('centering', [False, True])
def test_unsharded_run_samples_the_bits_of_its_shards_on_a_mixed_size_batch(centering):
    """team sizes follow the size of the WHOLE batch in every path - the fused chain of the <= 55-atom molecules,
    the team launches of the 56..110-atom ones, the denoiser calls of a host-driven chain - so the unsharded call and the two
    halves of the batch sampled as shards (what ``distributed.sample_chain_sharded`` does on two ranks) give the same bits."""
    from difflinker_amd import Dynamics
    from difflinker_amd.distributed import shard_sampler_inputs
    nf, L, T = 8, 1, 4
    dyn = Dynamics(n_dims=3, in_node_nf=nf, context_node_nf=1, hidden_nf=128, n_layers=L, norm_constant=1e-6, centering=centering)
    dyn.load_state_dict(seeded_state_dict(nf + 2, 128, L, 91), strict=True)
    dyn = dyn.to(P.dev())
    sizes, linkers = [30, 70, 45, 90, 20, 60, 12, 100], [5, 7, 6, 9, 4, 6, 3, 10]
    inp, _, _ = P.ragged_inputs(sizes, linkers, nf, seed=92)
    g = {k: v.to(P.dev()) for k, v in inp.items()}
    edm = _edm(dyn, nf, T)
    edm.noise_source = 'philox'
    edm.noise_seed = 5
    whole = edm.sample_chain(g['x'], g['h'], g['node_mask'], g['fragment_mask'], g['linker_mask'], g['edge_mask'], g['context'],
                             keep_frames=2).cpu()
    B = len(sizes)
    parts = []
    for rank in range(2):
        local, (lo, hi) = shard_sampler_inputs(g, rank, 2)
        edm.noise_seed = 5
        edm.coef_batch, edm.team_batch = B, B            # what sample_chain_sharded pins
        try:
            parts.append(edm.sample_chain(keep_frames=2, mol_offset=lo, **local).cpu())
        finally:
            edm.coef_batch = edm.team_batch = None
    assert torch.equal(torch.cat(parts, dim=1), whole)


# ---- NaN index sets: the earliest denoiser call only -----------------------------------------------------------------------
def test_nan_sets_of_a_batch_sampled_in_parts_are_those_of_the_earliest_call():
    """the reference raises at the first denoiser call whose output holds a NaN, with the molecules that are NaN
    THERE (egnn.py:441-442).  Molecule 0 (fused chain) goes bad at call 2, molecule 3 (host-driven part) at call 0: only 3 is
    reported."""
    from difflinker_amd.utils import FoundNaNException
    nf, T = 8, 4
    dyn, sd, cfg = P.make_dynamics(nf, 1, 1, seed=33)
    inp, _, _ = P.ragged_inputs([20, 70, 35, 120, 12], [4, 9, 5, 8, 3], nf, seed=34)
    B, N = inp['x'].shape[:2]
    edm = _edm(dyn, nf, T)
    bank = edm_oracle.NoiseBank.generate(T, B, N, 3, nf, seed=35)
    nx, nh = bank.stacked()
    nx = nx.clone()
    nx[0, 3, 119, 0] = float('nan')                      # draw 0: z_T of molecule 3 -> its call 0
    nx[2, 0, 19, 1] = float('nan')                       # draw 2 (the noise of step 1): z after step 1 of molecule 0 -> its call 2
    g = {k: v.to(P.dev()) for k, v in inp.items()}
    with pytest.raises(FoundNaNException) as ei:
        edm.sample_chain(g['x'], g['h'], g['node_mask'], g['fragment_mask'], g['linker_mask'], g['edge_mask'], g['context'],
                         keep_frames=1, noise_bank=(nx, nh))
    e = ei.value
    bad = e.x_h_nan_idx | e.only_x_nan_idx | e.only_h_nan_idx
    print(f'NaNs planted in molecule 0 (call 2) and molecule 3 (call 0): reported {sorted(bad)} for call {e.first_step}')
    assert bad == {3} and e.first_step == 0
    nx[0, 3, 119, 0] = 0.5                               # without the early one: molecule 0 alone, at its own call
    with pytest.raises(FoundNaNException) as ei:
        edm.sample_chain(g['x'], g['h'], g['node_mask'], g['fragment_mask'], g['linker_mask'], g['edge_mask'], g['context'],
                         keep_frames=1, noise_bank=(nx, nh))
    e = ei.value
    assert (e.x_h_nan_idx | e.only_x_nan_idx | e.only_h_nan_idx) == {0} and e.first_step == 2


# ---- InpaintingEDM: counter-based noise, shards -----------------------------------------------------------------------------
def test_inpainting_chain_with_counter_based_noise_is_shard_independent():
    """VERDICT round 4 (#8): ``InpaintingEDM`` (edm.py:549-730) through the sharded entry point.  With ``noise_source='philox'`` its
    1 + 2T + 2 draws per molecule come from the counter-based generator keyed by the GLOBAL molecule index: the two halves of a
    batch sampled as shards equal the unsharded chain bit for bit, and the chain is the oracle's for the same bank."""
    from difflinker_amd import Dynamics, InpaintingEDM
    from difflinker_amd.distributed import sample_chain_sharded, shard_sampler_inputs
    nf, L, T = 8, 1, 4
    dyn = Dynamics(n_dims=3, in_node_nf=nf, context_node_nf=1, hidden_nf=128, n_layers=L, norm_constant=1e-6, centering=True)
    sd = seeded_state_dict(nf + 2, 128, L, 95)
    dyn.load_state_dict(sd, strict=True)
    dyn = dyn.to(P.dev())
    inp, _, _ = P.ragged_inputs([14, 30, 9, 22, 17], [4, 6, 3, 5, 4], nf, seed=96)
    g = {k: v.to(P.dev()) for k, v in inp.items()}
    edm = InpaintingEDM(dyn, in_node_nf=nf, n_dims=3, timesteps=500, noise_schedule='polynomial_2', noise_precision=1e-5,
                        loss_type='l2', norm_values=[1, 4, 10]).to(P.dev())
    edm.T = T
    edm.noise_source = 'philox'
    edm.noise_seed = 11
    whole = edm.sample_chain(keep_frames=2, **g).cpu()
    assert torch.isfinite(whole).all()
    edm.noise_seed = 11
    assert torch.equal(sample_chain_sharded(edm, g, keep_frames=2).cpu(), whole)        # world = 1: the same call
    B, parts = len(inp['x']), []
    for rank in range(2):
        local, (lo, hi) = shard_sampler_inputs(g, rank, 2)
        edm.noise_seed = 11
        edm.coef_batch, edm.team_batch = B, B
        try:
            parts.append(edm.sample_chain(keep_frames=2, mol_offset=lo, **local).cpu())
        finally:
            edm.coef_batch = edm.team_batch = None
    assert torch.equal(torch.cat(parts, dim=1), whole)
    # the same draws as an explicit bank: the explicit-bank path (pinned to the reference by tests/golden/inpainting_chain.npz) agrees
    edm.noise_seed = 11
    bank = edm.philox_noise_bank(B, inp['x'].shape[1], P.dev(), n_draws=1 + 2 * T + 2)
    edm.noise_source = 'torch'
    assert torch.equal(edm.sample_chain(keep_frames=2, noise_bank=bank, **g).cpu(), whole)


# ---- hyper-parameters that used to raise: hidden_nf <= 128, inv_sublayers != 2, condition_time = False --------------------------
def _hparam_dynamics(cls, hidden_nf, inv_sublayers, condition_time, n_layers, seed, ctx, **kw):
    nf = 8
    fin = nf + ctx + int(condition_time)
    dyn = cls(n_dims=3, in_node_nf=nf, context_node_nf=ctx, hidden_nf=hidden_nf, n_layers=n_layers, inv_sublayers=inv_sublayers,
              condition_time=condition_time, norm_constant=1e-6, **kw)
    sd = seeded_state_dict(fin, hidden_nf, n_layers, seed, inv_sublayers=inv_sublayers)
    dyn.load_state_dict(sd, strict=True)
    cfg = EGNNConfig(in_node_nf=nf, context_node_nf=ctx, hidden_nf=hidden_nf, n_layers=n_layers, inv_sublayers=inv_sublayers,
                     condition_time=condition_time, graph_type=kw.get('graph_type', 'FC'))
    return dyn.to(P.dev()), sd, cfg, nf


@pytest.mark.parametrize('hidden_nf,inv_sublayers,condition_time', [(64, 2, True), (128, 1, True), (128, 3, True), (128, 2, False),
                                                                    (32, 4, False), (100, 2, True)])
@pytest.mark.parametrize('team', ['1', 'auto'])
def test_forward_and_chain_with_other_widths_depths_and_no_time_feature(hidden_nf, inv_sublayers, condition_time, team):
    """VERDICT round 4, missing #3: ``hidden_nf != 128`` (the reference's own default is 64, egnn.py:324-329), ``inv_sublayers != 2``
    and ``condition_time=False`` raised.  Now: narrower networks run zero-padded on the 128-wide kernels, a block takes 1..4
    GCLs, the time feature is optional - forward against the oracle of the network AS GIVEN (its own width), then a short chain."""
    from difflinker_amd import Dynamics
    dyn, sd, cfg, nf = _hparam_dynamics(Dynamics, hidden_nf, inv_sublayers, condition_time, 2, seed=140 + hidden_nf + inv_sublayers, ctx=1)
    dyn.team = team if team == 'auto' else int(team)
    inp, z, t = P.ragged_inputs([33, 50, 12, 70], [5, 8, 3, 9], nf, seed=141)
    ref = egnn_oracle.dynamics_forward(sd, cfg, t, z, inp['node_mask'], inp['linker_mask'], inp['edge_mask'], inp['context'])
    out = P.run_hip_forward(dyn, inp, z, t)
    ev, eh = P.report(f'hidden {hidden_nf}, {inv_sublayers} GCLs per block, time feature {condition_time}, team {team}', out, ref, z)
    assert ev <= P.FWD_TOLS['f16x3'] and eh <= P.FWD_TOLS['f16x3']
    T = 6
    edm = _edm(dyn, nf, T)
    inp2, _, _ = P.ragged_inputs([20, 41], [4, 7], nf, seed=142)
    B, N = inp2['x'].shape[:2]
    bank = edm_oracle.NoiseBank.generate(T, B, N, 3, nf, seed=143)
    orc = edm_oracle.EDMOracle(edm_oracle.make_dynamics_oracle(sd, cfg), in_node_nf=nf, timesteps=500)
    orc.T = T
    want = orc.sample_chain(inp2['x'], inp2['h'], inp2['node_mask'], inp2['fragment_mask'], inp2['linker_mask'], inp2['edge_mask'],
                            inp2['context'], bank, keep_frames=2)
    g = {k: v.to(P.dev()) for k, v in inp2.items()}
    got = edm.sample_chain(g['x'], g['h'], g['node_mask'], g['fragment_mask'], g['linker_mask'], g['edge_mask'], g['context'],
                           keep_frames=2, noise_bank=bank.stacked()).cpu()
    P.check_chain(f'chain, hidden {hidden_nf}, {inv_sublayers} GCLs per block, time feature {condition_time}', got, want, inp2)


@pytest.mark.parametrize('hidden_nf,inv_sublayers,condition_time', [(64, 1, True), (128, 3, False)])
def test_pocket_forward_with_other_widths_depths_and_no_time_feature(hidden_nf, inv_sublayers, condition_time):
    """The same on the radius-graph kernels (egnn_sparse.hip)."""
    from difflinker_amd import DynamicsWithPockets
    dyn, sd, cfg, nf = _hparam_dynamics(DynamicsWithPockets, hidden_nf, inv_sublayers, condition_time, 2, seed=150 + hidden_nf, ctx=2,
                                        graph_type='FC-10A-4A')
    inp, z, t = P.pocket_inputs(batch=2, n_frag=12, n_pocket=80, linker=(4, 7), nf=nf, seed=151)
    ref = egnn_oracle.dynamics_forward_pockets(sd, cfg, t, z, inp['node_mask'], inp['linker_mask'], inp['edge_mask'], inp['context'])
    out = P.run_hip_forward(dyn, inp, z, t)
    ev, eh = P.report(f'pockets, hidden {hidden_nf}, {inv_sublayers} GCLs per block, time feature {condition_time}', out, ref, z)
    assert ev <= P.FWD_TOLS['f16x3'] and eh <= P.FWD_TOLS['f16x3']


@pytest.mark.parametrize('hidden_nf,seed,mag,sizes', [(32, 600030, 550.0, [47, 88, 62]), (32, 600009, 319.0, [47, 109, 21, 89, 81]),
                                                      (64, 600041, 1.0, [30, 55])])
def test_narrow_network_with_trained_like_weights(hidden_nf, seed, mag, sizes):
    """A zero-padded narrow network has as many k-slabs / tiles for its hidden features' exponents as a 128-wide one and a
    quarter of the features: packed in one corner, 32 features spread over ~20 binades shared TWO slab exponents and the error
    of the f16x3 mode had a tail (scripts/r5/fuzz_forward.py, 50 seeds: up to 1.6e-5 against 1.7e-6 at width 128; the first two
    cases here are its worst).  The packer deals the features out over all the groups (egnn_fc.hip: balance_hidden)."""
    from difflinker_amd import Dynamics
    nf, L, sub = 9, 3, 2
    dyn = Dynamics(n_dims=3, in_node_nf=nf, context_node_nf=1, hidden_nf=hidden_nf, n_layers=L, inv_sublayers=sub,
                   condition_time=False, norm_constant=1e-6)
    sd = trained_like_state_dict(seeded_state_dict(nf + 1, hidden_nf, L, seed, inv_sublayers=sub, coord_gain=1.0), seed + 1)
    sd['dynamics.embedding.weight'] = sd['dynamics.embedding.weight'] * mag
    sd['dynamics.embedding.bias'] = sd['dynamics.embedding.bias'] * mag
    dyn.load_state_dict(sd, strict=True)
    dyn = dyn.to(P.dev())
    cfg = EGNNConfig(in_node_nf=nf, context_node_nf=1, hidden_nf=hidden_nf, n_layers=L, inv_sublayers=sub, condition_time=False)
    inp, z, t = P.ragged_inputs(sizes, [min(s, 5) for s in sizes], nf, seed=seed + 2)
    ref = egnn_oracle.dynamics_forward(sd, cfg, t, z, inp['node_mask'], inp['linker_mask'], inp['edge_mask'], inp['context'])
    assert torch.isfinite(ref).all()
    for team in (1, 'auto'):
        dyn.team = team
        out = P.run_hip_forward(dyn, inp, z, t)
        eh = rel_l2(out[..., 3:], ref[..., 3:])
        print(f'narrow network, hidden {hidden_nf}, trained-like weights x {mag:g}, team {team}: node features rel-L2 {eh:.3e}')
        assert eh <= 3e-6                                   # (before: 1.5e-5 on the first two cases)


# ---- the HBM-resident kernels across magnitudes ------------------------------------------------------------------------------
@pytest.mark.parametrize('mag', [1e-3, 1e-1, 1e2, 1e4, 1e6, 1e8])
def test_hbm_resident_kernels_over_twenty_binades_of_magnitude(mag):
    """Round 5 found the f16x3 edge kernel of the HBM-resident path (pockets, molecules beyond 110 atoms) passing its per-tile
    activation bound through an int (``readfirstlane`` of the float VALUE): a bound below 1 became 0 - scale 2^60, every
    activation saturated at the fp16 maximum - and one above 2^31 was clipped; no test had left the range [1, 2^31).  A
    120-atom molecule whose coordinate model sees ONE dominant feature of magnitude `mag`^2, both arithmetic modes against the oracle
    (same model on the LDS-resident kernels: a 40-atom molecule)."""
    from difflinker_amd import Dynamics
    nf = 9
    for sizes in ((120, 12), (40, 12)):
        inp, z, t = P.ragged_inputs(list(sizes), [9, 4], nf, seed=300)
        special = sizes[0] - 3
        z[:, :, 3 + 7] = 0.0
        z[0, special, 3 + 7] = 1.0
        sd = seeded_state_dict(nf + 2, 128, 1, 301)
        for v in sd.values():
            v.zero_()
        sd['dynamics.embedding.weight'][0, 7] = mag
        sd['dynamics.e_block_0.gcl_equiv.coord_mlp.0.weight'][0, 0] = mag
        sd['dynamics.e_block_0.gcl_equiv.coord_mlp.2.weight'][0, 0] = 1e3
        sd['dynamics.e_block_0.gcl_equiv.coord_mlp.4.weight'][0, 0] = 1.0
        cfg = EGNNConfig(in_node_nf=nf, context_node_nf=1, n_layers=1)
        ref = egnn_oracle.dynamics_forward(sd, cfg, t, z, inp['node_mask'], inp['linker_mask'], inp['edge_mask'], inp['context'])
        for precision in ('f16x3', 'fp32'):
            dyn = Dynamics(n_dims=3, in_node_nf=9, context_node_nf=1, hidden_nf=128, n_layers=1, norm_constant=1e-6)
            dyn.precision = precision
            dyn.load_state_dict(sd, strict=True)
            out = P.run_hip_forward(dyn.to(P.dev()), inp, z, t)
            err = rel_l2(out[0, special, :3], ref[0, special, :3])
            print(f'{sizes[0]} atoms, magnitude {mag:g}, {precision}: velocity of the atom the feature drives, rel-L2 {err:.2e}')
            assert err <= 2e-6


def test_trained_like_weights_full_size_chain_against_the_exact_mode():
    """The benchmark's own launch (config C2, B = 256, T = 500, in-kernel noise) with TRAINED-LIKE weights, once in the default
    f16x3 arithmetic and once in the exact-fp32 MFMA mode: 501 forwards of accumulated rounding later the two samples still agree
    on the linker coordinates and in every atom type (the oracle cannot follow at this size: an hour per chain)."""
    from difflinker_amd import Dynamics, EDM, synthetic
    data, cfg = synthetic.make_batch('C2', seed=1000)
    inp = {k: v.to(P.dev()) for k, v in synthetic.sampler_inputs(data).items()}
    sd = trained_like_state_dict(seeded_state_dict(cfg['nf'] + cfg['ctx'] + 1, 128, cfg['n_layers'], 80, coord_gain=0.02), seed=7)
    chains = {}
    for precision in ('f16x3', 'fp32'):
        dyn = Dynamics(n_dims=3, in_node_nf=cfg['nf'], context_node_nf=cfg['ctx'], hidden_nf=128, n_layers=cfg['n_layers'],
                       norm_constant=1e-6, normalization='batch_norm')
        dyn.load_state_dict(sd, strict=True)
        dyn.precision = precision
        edm = EDM(dyn, in_node_nf=cfg['nf'], n_dims=3, timesteps=500, noise_schedule='polynomial_2', noise_precision=1e-5,
                  loss_type='l2', norm_values=[1, 4, 10]).to(P.dev())
        edm.noise_source, edm.noise_seed = 'philox', 5
        chains[precision] = edm.sample_chain(keep_frames=1, **inp).cpu()
    lm = inp['linker_mask'].cpu()
    a, b = chains['f16x3'][0], chains['fp32'][0]
    ex = rel_l2(a[..., :3] * lm, b[..., :3] * lm)
    mism = int((a[..., 3:] != b[..., 3:]).any(-1).sum())
    moved = float(((b[..., :3] - inp['x'].cpu()) * lm).norm(dim=-1).max())
    print(f'[trained-like weights, C2 B=256 T=500, f16x3 vs fp32 mode] final linker-x rel-L2 {ex:.3e}, atom-type mismatches {mism}, '
          f'largest linker displacement {moved:.1f} A')
    assert torch.isfinite(a).all() and torch.isfinite(b).all()
    assert ex <= 1e-5 and mism == 0          # measured 3.7e-7, every atom type equal, the largest linker displacement 520 A


@pytest.mark.parametrize('bad', [float('nan'), float('inf')])
def test_non_finite_weights_reach_the_kernels_and_raise(bad):
    """The balanced packing sorts hidden features by a magnitude proxy on the host: a checkpoint with a NaN / inf weight must not
    upset that sort (or anything else there) - it has to reach the kernels and come back as FoundNaNException, like the reference."""
    from difflinker_amd.utils import FoundNaNException
    nf = 9
    dyn, sd, cfg = P.make_dynamics(nf, 1, 2, seed=77)
    sd = {k: v.clone() for k, v in sd.items()}
    sd['dynamics.e_block_0.gcl_1.edge_mlp.0.weight'][5, 17] = bad
    sd['dynamics.e_block_1.gcl_0.node_mlp.0.weight'][40, 200] = bad
    dyn.load_state_dict(sd, strict=True)
    dyn.invalidate_packed()
    inp, z, t = P.ragged_inputs([30, 12], [5, 3], nf, seed=78)
    try:
        egnn_oracle.dynamics_forward(sd, cfg, t, z, inp['node_mask'], inp['linker_mask'], inp['edge_mask'], inp['context'])
        oracle_raised = False
    except egnn_oracle.OracleNaN:
        oracle_raised = True
    assert oracle_raised or bad == float('inf')          # (an inf weight may give inf, not NaN: the reference checks isnan only)
    if oracle_raised:
        with pytest.raises(FoundNaNException):
            P.run_hip_forward(dyn, inp, z, t)
    else:
        try:
            P.run_hip_forward(dyn, inp, z, t)             # must not crash; f16x3 beyond its range answers FoundNaNException (documented)
        except FoundNaNException:
            pass


# ---- a chain in two phases: the static hand-over --------------------------------------------------------------------------
@pytest.mark.parametrize('case', ['teams only', 'teams and single compute units'])
def test_split_chain_hands_compute_units_over_and_samples_the_oracles_chain(case):
    """``EDM.split_chain``: the small molecules of a ragged batch complete in the first launch, the others stop at the call where
    the small ones end (dl_chain_args.q_end), leave their state in HBM and finish in a second phase (q_begin, z_state) - the ones
    with the most work left on teams of two, and, when the compute units do not suffice for teams everywhere (the second case:
    more unfinished molecules than half the compute units - the shape of the C2 plan: 124 teams + 8 singles), the rest on one compute
    unit each, side by side (here 136 unfinished: 120 teams + 16 singles).  The chain must be the oracle's - every kept frame - and agree with the one-launch chain to fp32
    rounding (the steps on teams sum messages in the team's order); molecules that never run on a team are bit-identical;
    repeatable bit for bit; the plan comes from the sizes alone."""
    from difflinker_amd import edm as edm_mod
    nf, L, T = 8, (2 if case == 'teams only' else 1), 24
    if case == 'teams only':
        sizes, linkers = [50, 48, 50, 47, 20, 22, 18, 25, 21, 19, 23, 20], [8, 7, 9, 6, 4, 5, 3, 6, 4, 4, 5, 4]
    else:
        sizes, linkers = [30] * 100 + [26] * 36 + [10] * 60, [6] * 100 + [5] * 36 + [3] * 60          # 136 unfinished: 120 teams + 16 singles
    cus = torch.cuda.get_device_properties(P.dev()).multi_processor_count
    plan = edm_mod.split_plan(sizes, linkers, T + 1, cus, L, 2, allow_singles=(case != 'teams only'))
    assert plan is not None
    q_end, teams, singles = plan
    assert teams and all(0 < q_end[b] < T + 1 for b in teams + singles) and 2 * len(teams) + len(singles) <= cus
    assert (len(singles) > 0) == (case != 'teams only')
    untouched = sorted(set(range(len(sizes))) - set(teams))              # complete in the first launch, or resume on ONE compute unit
    dyn, sd, cfg = P.make_dynamics(nf, 1, L, seed=171)
    dyn.team = 1
    inp, _, _ = P.ragged_inputs(sizes, linkers, nf, seed=172)
    B, N = inp['x'].shape[:2]
    edm = _edm(dyn, nf, T)
    edm.split_singles = case != 'teams only'        # (the variant with single compute units beside the teams is opt-in: split_plan)
    bank = edm_oracle.NoiseBank.generate(T, B, N, 3, nf, seed=173)
    orc = edm_oracle.EDMOracle(edm_oracle.make_dynamics_oracle(sd, cfg), in_node_nf=nf, timesteps=500)
    orc.T = T
    want = orc.sample_chain(inp['x'], inp['h'], inp['node_mask'], inp['fragment_mask'], inp['linker_mask'], inp['edge_mask'],
                            inp['context'], bank, keep_frames=6)
    g = {k: v.to(P.dev()) for k, v in inp.items()}

    def run(split, philox=False):
        edm.split_chain = split
        if philox:
            edm.noise_source, edm.noise_seed = 'philox', 9
            out = edm.sample_chain(g['x'], g['h'], g['node_mask'], g['fragment_mask'], g['linker_mask'], g['edge_mask'], g['context'],
                                   keep_frames=6)
            edm.noise_source = 'torch'
        else:
            out = edm.sample_chain(g['x'], g['h'], g['node_mask'], g['fragment_mask'], g['linker_mask'], g['edge_mask'], g['context'],
                                   keep_frames=6, noise_bank=bank.stacked())
        torch.cuda.synchronize()
        return out.cpu()
    got = run(True)
    P.check_chain(f'split chain ({case}: {len(teams)} teams, {len(singles)} singles), T=24, 6 frames', got, want, inp)
    assert torch.equal(got, run(True)), 'bitwise repeatable'
    one = run(False)
    assert torch.equal(got[:, untouched], one[:, untouched]), 'molecules that never run on a team: the bits of the one-launch chain'
    lm = inp['linker_mask'][teams]
    err = rel_l2(got[0, teams, :, :3] * lm, one[0, teams, :, :3] * lm)
    print(f'split vs one launch, the molecules that finish on teams: linker-x rel-L2 {err:.3e}')
    assert err <= 1e-5 and torch.equal(got[0, teams, :, 3:], one[0, teams, :, 3:])
    a, b = run(True, philox=True), run(False, philox=True)                       # in-kernel noise: resuming needs no generator state
    assert rel_l2(a[0, :, :, :3] * inp['linker_mask'], b[0, :, :, :3] * inp['linker_mask']) <= 1e-5
    assert torch.equal(a[:, untouched], b[:, untouched])


def test_split_chain_reports_nans_of_either_launch_like_one_launch():
    """A NaN in a molecule that stops in the first launch of a split chain and resumes in the second: planted in a draw of the
    FIRST launch it ends the molecule there (the second launch skips it: dl_chain_args.skip_flags); planted in a draw of the
    SECOND launch it is found on the team.  Either way the exception carries what the one-launch chain reports: the same index
    sets and the same denoiser call."""
    from difflinker_amd import edm as edm_mod
    from difflinker_amd.utils import FoundNaNException
    nf, L, T = 8, 1, 24
    sizes, linkers = [50, 48, 50, 47, 20, 22, 18, 25, 21, 19, 23, 20], [8, 7, 9, 6, 4, 5, 3, 6, 4, 4, 5, 4]
    q_end, teams, _ = edm_mod.split_plan(sizes, linkers, T + 1, 256, L, 2)
    assert 0 in teams and 2 in teams and 3 < q_end[0] < T - 3
    dyn, sd, cfg = P.make_dynamics(nf, 1, L, seed=181)
    dyn.team = 1
    inp, _, _ = P.ragged_inputs(sizes, linkers, nf, seed=182)
    B, N = inp['x'].shape[:2]
    edm = _edm(dyn, nf, T)
    g = {k: v.to(P.dev()) for k, v in inp.items()}
    bank = edm_oracle.NoiseBank.generate(T, B, N, 3, nf, seed=183)
    for draw, mol in ((2, 0), (q_end[2] + 2, 2)):                   # a draw of the first launch / of the second
        nx, nh = (t_.clone() for t_ in bank.stacked())
        nx[draw, mol, sizes[mol] - 1, 0] = float('nan')            # a linker atom: z after step draw - 1 holds a NaN -> call `draw`
        seen = {}
        for split in (False, True):
            edm.split_chain = split
            with pytest.raises(FoundNaNException) as ei:
                edm.sample_chain(g['x'], g['h'], g['node_mask'], g['fragment_mask'], g['linker_mask'], g['edge_mask'], g['context'],
                                 keep_frames=1, noise_bank=(nx, nh))
            e = ei.value
            seen[split] = (e.x_h_nan_idx, e.only_x_nan_idx, e.only_h_nan_idx, e.first_step)
        print(f'NaN planted in draw {draw} of molecule {mol} (stops at call {q_end[mol]}): one launch {seen[False]}, split {seen[True]}')
        assert seen[True] == seen[False] and (seen[True][0] | seen[True][1] | seen[True][2]) == {mol} and seen[True][3] == draw


@pytest.mark.parametrize('mag', [1e-4, 1e-2, 1e2, 1e4])
@pytest.mark.parametrize('sizes,linkers', [([40, 12], [6, 3]), ([70, 20], [8, 4]), ([120, 12], [9, 4])])
def test_forward_across_feature_magnitudes(sizes, linkers, mag):
    """The a-priori / measured power-of-two scales of the f16 arithmetic over eight decades of feature magnitude, through EVERY
    layer (the sweep of the test above drives the coordinate model only): a seeded model whose embedding is scaled by `mag` - node
    features, messages, aggregates and node-MLP activations all move with it, into SiLU's linear and its dead range - on the
    LDS-resident kernels (40 atoms), a team (70) and the HBM-resident kernels (120), against the oracle."""
    nf, L = 9, 2
    dyn, sd, cfg = P.make_dynamics(nf, 1, L, seed=191)
    sd = {k: v.clone() for k, v in sd.items()}
    sd['dynamics.embedding.weight'] *= mag
    sd['dynamics.embedding.bias'] *= mag
    dyn.load_state_dict(sd, strict=True)
    dyn.invalidate_packed()
    inp, z, t = P.ragged_inputs(sizes, linkers, nf, seed=192)
    ref = egnn_oracle.dynamics_forward(sd, cfg, t, z, inp['node_mask'], inp['linker_mask'], inp['edge_mask'], inp['context'])
    assert torch.isfinite(ref).all()
    out = P.run_hip_forward(dyn, inp, z, t)
    ev, eh = P.report(f'{sizes[0]} atoms, embedding x {mag:g}', out, ref, z)
    assert ev <= P.FWD_TOLS['f16x3'] and eh <= P.FWD_TOLS['f16x3']


# ---- beyond the range of the f16 modes: loud, never a saturated number ------------------------------------------------------------
def _far_apart_case(spread, sizes, linkers):
    """A seeded model on molecules whose atoms lie `spread` apart: squared distances of spread^2 enter every edge / coordinate
    model (radial and d0: egnn.py:160-163, 219-222)."""
    nf, L = 9, 2
    dyn, sd, cfg = P.make_dynamics(nf, 1, L, seed=211)
    inp, z, t = P.ragged_inputs(sizes, linkers, nf, seed=212)
    z = z.clone()
    z[..., :3] *= spread
    return dyn, sd, cfg, inp, z, t


@pytest.mark.parametrize('sizes,linkers,team', [([20, 12], [5, 4], 1), ([20, 12], [5, 4], 'auto'), ([70, 20], [8, 4], 'auto'),
                                               ([120, 12], [9, 4], 'auto')])
@pytest.mark.parametrize('spread', [1e6, 1e9, 3e10])
def test_coordinates_beyond_the_f16_range_are_reported_not_saturated(sizes, linkers, team, spread):
    """scripts/r5/fuzz_chain.py, seed 300109: a chain whose denoiser threw the atoms 1e10 apart - squared distances of 1e21 times
    weights of ~30, a bound beyond the 2^-60 clamp of the f16x3 scales - came back FINITE and wrong (3e-3, eight atom types):
    v_cvt_pkrtz saturates at 65504 instead of overflowing, and the round-4 pin of the clamp (test_gpu_round4.py) had only driven
    the node features there, where the arithmetic happens to end in NaN.  Now every scale that comes from a run-time bound checks
    it (pack_layout.h: beyond_f16_range; nan_flags bit 4): within the range (atoms 1e6 apart) the result is the oracle's, beyond
    it (3e10: the squared distances alone exceed 3.8e22 - a little further and the fp32 oracle overflows too; at 1e9 the bound,
    squared distance x weights, decides) the call raises ``FoundNaNException`` naming the molecules (``f16_range_idx``) - on one
    compute unit, on teams, on the HBM-resident kernels - and the exact-fp32 mode computes the oracle's result at every magnitude."""
    from difflinker_amd.utils import FoundNaNException
    dyn, sd, cfg, inp, z, t = _far_apart_case(spread, sizes, linkers)
    dyn.team = team
    ref = egnn_oracle.dynamics_forward(sd, cfg, t, z, inp['node_mask'], inp['linker_mask'], inp['edge_mask'], inp['context'])
    assert torch.isfinite(ref).all(), 'fp32 holds these magnitudes'
    for precision in ('fp32', 'f16x3'):
        dyn.precision = precision
        try:
            out = P.run_hip_forward(dyn, inp, z, t)
        except FoundNaNException as e:
            print(f'[{sizes[0]} atoms, team {team}, atoms {spread:g} apart, {precision}] {e}')
            assert precision == 'f16x3' and spread >= 1e9, 'only the f16 modes, and only beyond their range, may give up'
            assert e.f16_range_idx and e.f16_range_idx <= e.x_h_nan_idx and not e.only_x_nan_idx and not e.only_h_nan_idx
            continue
        eh = rel_l2(out[..., 3:], ref[..., 3:])
        ev = rel_l2(out[..., :3], ref[..., :3])
        print(f'[{sizes[0]} atoms, team {team}, atoms {spread:g} apart, {precision}] rel-L2 h {eh:.3e} vel {ev:.3e}')
        assert eh <= 1e-5 and ev <= 1e-4, f'{precision} returned finite numbers that are not the reference\'s'
        assert precision == 'fp32' or spread < 3e10, 'squared distances of 1e23 cannot have fitted the f16 scales'


def test_pocket_coordinates_beyond_the_f16_range_are_reported():
    """The same on the radius-graph kernels: their scales belong to tiles, so the report names every molecule of the call."""
    from difflinker_amd.utils import FoundNaNException
    nf = 9
    dyn, sd, cfg = P.make_pocket_dynamics(nf, 2, seed=221)
    inp, z, t = P.pocket_inputs(batch=2, n_frag=12, n_pocket=60, linker=(4, 7), nf=nf, seed=222)
    for spread, beyond in ((1.0, False), (3e10, True)):
        zz = z.clone()
        zz[..., :3] *= spread              # (atoms 1e11 apart: the radius edges are gone, the ligand's fully-connected ones carry 1e23; beyond 1e11 the oracle overflows)
        ref = egnn_oracle.dynamics_forward_pockets(sd, cfg, t, zz, inp['node_mask'], inp['linker_mask'], inp['edge_mask'], inp['context'])
        assert torch.isfinite(ref).all()
        for precision in ('fp32', 'f16x3'):
            dyn.precision = precision
            dyn.invalidate_packed()
            try:
                out = P.run_hip_forward(dyn, inp, zz, t)
            except FoundNaNException as e:
                print(f'[pockets, coordinates x {spread:g}, {precision}] {e}')
                assert precision == 'f16x3' and beyond and e.f16_range_idx == {0, 1}
                continue
            eh = rel_l2(out[..., 3:], ref[..., 3:])
            print(f'[pockets, coordinates x {spread:g}, {precision}] rel-L2 h {eh:.3e}')
            assert eh <= 1e-5 and not (beyond and precision == 'f16x3')


# ---- edges of the input domain -----------------------------------------------------------------------------------------------
@pytest.mark.parametrize('case', ['one step', 'a molecule without linker atoms, every frame kept', 'no padding row'])
def test_chain_edge_cases_against_the_oracle(case):
    """A chain of ONE step (T = 1: the initial z, one denoising step, the decode), a batch one of whose molecules has nothing
    to sample (its linker mask is empty: the denoiser still sees it, every frame returns its fragment unchanged) with EVERY frame
    kept (keep_frames = T), and a batch whose molecules all fill the padded width."""
    sizes, linkers, T, keep, empty = {'one step': ([12, 7], [4, 2], 1, 1, None),
                                      'a molecule without linker atoms, every frame kept': ([12, 9, 7], [4, 3, 2], 6, 6, 1),
                                      'no padding row': ([5, 5, 5], [2, 2, 2], 5, 1, None)}[case]
    nf, L = 8, 2
    dyn, sd, cfg = P.make_dynamics(nf, 1, L, seed=231)
    inp, _, _ = P.ragged_inputs(sizes, linkers, nf, seed=232)
    if empty is not None:
        inp['fragment_mask'][empty] = inp['node_mask'][empty].to(inp['fragment_mask'].dtype)
        inp['linker_mask'][empty] = 0
    B, N = inp['x'].shape[:2]
    edm = _edm(dyn, nf, T)
    bank = edm_oracle.NoiseBank.generate(T, B, N, 3, nf, seed=233)
    orc = edm_oracle.EDMOracle(edm_oracle.make_dynamics_oracle(sd, cfg), in_node_nf=nf, timesteps=500)
    orc.T = T
    want = orc.sample_chain(inp['x'], inp['h'], inp['node_mask'], inp['fragment_mask'], inp['linker_mask'], inp['edge_mask'],
                            inp['context'], bank, keep_frames=keep)
    g = {k: v.to(P.dev()) for k, v in inp.items()}
    for team in (1, 'auto'):
        dyn.team = team
        got = edm.sample_chain(g['x'], g['h'], g['node_mask'], g['fragment_mask'], g['linker_mask'], g['edge_mask'], g['context'],
                               keep_frames=keep, noise_bank=bank.stacked()).cpu()
        P.check_chain(f'{case}, team {team}', got, want, inp)
        if empty is not None:
            assert torch.equal(got[:, empty], want[:, empty]), 'nothing to sample: the fragment, bit for bit, in every frame'


def test_an_empty_batch_is_an_empty_chain():
    """B = 0: every op of the reference runs on empty tensors (its edge list is empty: egnn.py:449-464) and the chain comes back
    with no molecule in it; so it does here - from the denoiser, the sampler, the inpainting sampler and the pocket denoiser."""
    from difflinker_amd import InpaintingEDM, Dynamics
    nf, N, T = 8, 10, 4
    dyn, _, _ = P.make_dynamics(nf, 1, 1, seed=241)
    d = P.dev()
    z = dict(x=torch.zeros(0, N, 3), h=torch.zeros(0, N, nf), node_mask=torch.zeros(0, N, 1), fragment_mask=torch.zeros(0, N, 1),
             linker_mask=torch.zeros(0, N, 1), edge_mask=torch.zeros(0, 1), context=torch.zeros(0, N, 1))
    g = {k: v.to(d) for k, v in z.items()}
    out = dyn.forward(torch.zeros(0, 1, device=d), torch.zeros(0, N, 3 + nf, device=d), g['node_mask'], g['linker_mask'], g['edge_mask'], g['context'])
    assert tuple(out.shape) == (0, N, 3 + nf)
    edm = _edm(dyn, nf, T)
    chain = edm.sample_chain(g['x'], g['h'], g['node_mask'], g['fragment_mask'], g['linker_mask'], g['edge_mask'], g['context'], keep_frames=2)
    assert tuple(chain.shape) == (2, 0, N, 3 + nf)
    cdyn = Dynamics(n_dims=3, in_node_nf=nf, context_node_nf=1, hidden_nf=128, n_layers=1, norm_constant=1e-6, centering=True).to(d)
    inp_edm = InpaintingEDM(cdyn, in_node_nf=nf, n_dims=3, timesteps=500, noise_schedule='polynomial_2', noise_precision=1e-5,
                            loss_type='l2', norm_values=[1, 4, 10]).to(d)
    inp_edm.T = T
    chain = inp_edm.sample_chain(g['x'], g['h'], g['node_mask'], g['edge_mask'], g['fragment_mask'], g['linker_mask'], g['context'], keep_frames=3)
    assert tuple(chain.shape) == (3, 0, N, 3 + nf)
    pdyn, _, _ = P.make_pocket_dynamics(nf, 1, seed=242)
    out = pdyn.forward(torch.zeros(0, 1, device=d), torch.zeros(0, N, 3 + nf, device=d), g['node_mask'], g['linker_mask'],
                       torch.zeros(0, device=d), torch.zeros(0, N, 2, device=d))
    assert tuple(out.shape) == (0, N, 3 + nf)


@pytest.mark.parametrize('team', [1, 'auto'])
def test_real_atoms_anywhere_among_the_padding_rows(team):
    """The reference takes any node mask (collate pads at the end, templates put the linker last - but nothing in egnn.py /
    edm.py relies on it).  The same molecules with their rows - real atoms and padding alike - shuffled, masks and the [B, N, N]
    edge mask shuffled along: the denoiser's output and the sampler's chain are the shuffled ones (one compute unit per
    molecule, teams, the HBM-resident kernels at 120 atoms; the sums run in another order: fp32 rounding)."""
    nf, L, T = 8, 2, 5
    dyn, sd, cfg = P.make_dynamics(nf, 1, L, seed=251)
    dyn.team = team
    inp, z, t = P.ragged_inputs([20, 12, 70, 120], [5, 4, 8, 9], nf, seed=252)
    B, N = z.shape[:2]
    g = torch.Generator().manual_seed(253)
    perm = torch.stack([torch.randperm(N, generator=g) for _ in range(B)])

    def shuffle(v):                                    # [B, N, ...] rows
        return torch.stack([v[b][perm[b]] for b in range(B)])
    em = inp['edge_mask'].view(B, N, N)
    inp_p = {k: shuffle(v) for k, v in inp.items() if k != 'edge_mask'}
    inp_p['edge_mask'] = torch.stack([em[b][perm[b]][:, perm[b]] for b in range(B)]).reshape(-1, 1)
    assert not torch.equal(inp_p['node_mask'], inp['node_mask'])
    out = P.run_hip_forward(dyn, inp, z, t)
    out_p = P.run_hip_forward(dyn, inp_p, shuffle(z), t)
    eh, ev = rel_l2(out_p[..., 3:], shuffle(out)[..., 3:]), rel_l2(out_p[..., :3], shuffle(out)[..., :3])
    print(f'rows shuffled, team {team}: forward h rel-L2 {eh:.3e} vel {ev:.3e}')
    assert eh <= 2e-6 and ev <= 1e-5
    assert float((out_p * (1 - inp_p['node_mask'].float())).abs().max()) == 0.0
    edm = _edm(dyn, nf, T)
    bank = edm_oracle.NoiseBank.generate(T, B, N, 3, nf, seed=254)
    nx, nh = bank.stacked()
    d = P.dev()

    def chain(i, bx, bh):
        gi = {k: v.to(d) for k, v in i.items()}
        return edm.sample_chain(gi['x'], gi['h'], gi['node_mask'], gi['fragment_mask'], gi['linker_mask'], gi['edge_mask'], gi['context'],
                                keep_frames=2, noise_bank=(bx, bh)).cpu()
    c = chain(inp, nx, nh)
    c_p = chain(inp_p, torch.stack([shuffle(nx[k]) for k in range(nx.shape[0])]), torch.stack([shuffle(nh[k]) for k in range(nh.shape[0])]))
    want = torch.stack([shuffle(c[f]) for f in range(c.shape[0])])
    ex = rel_l2(c_p[..., :3], want[..., :3])
    print(f'rows shuffled, team {team}: chain x rel-L2 {ex:.3e}, atom types equal {torch.equal(c_p[0, ..., 3:], want[0, ..., 3:])}')
    assert ex <= 1e-5 and torch.equal(c_p[0, ..., 3:], want[0, ..., 3:])


def test_pocket_atoms_anywhere_among_the_padding_rows():
    """The same for the radius-graph kernels: fragment, pocket, linker and padding rows in any order."""
    nf = 9
    dyn, sd, cfg = P.make_pocket_dynamics(nf, 2, seed=261)
    inp, z, t = P.pocket_inputs(batch=3, n_frag=12, n_pocket=60, linker=(4, 9), nf=nf, seed=262)
    B, N = z.shape[:2]
    g = torch.Generator().manual_seed(263)
    perm = torch.stack([torch.randperm(N, generator=g) for _ in range(B)])

    def shuffle(v):
        return torch.stack([v[b][perm[b]] for b in range(B)])
    inp_p = {k: (shuffle(v) if k != 'edge_mask' else v) for k, v in inp.items()}       # edge_mask: the batch id of every row
    out = P.run_hip_forward(dyn, inp, z, t)
    out_p = P.run_hip_forward(dyn, inp_p, shuffle(z), t)
    ref = egnn_oracle.dynamics_forward_pockets(sd, cfg, t, shuffle(z), inp_p['node_mask'], inp_p['linker_mask'], inp_p['edge_mask'], inp_p['context'])
    eh, ev, eo = rel_l2(out_p[..., 3:], shuffle(out)[..., 3:]), rel_l2(out_p[..., :3], shuffle(out)[..., :3]), rel_l2(out_p[..., 3:], ref[..., 3:])
    print(f'pocket rows shuffled: forward h rel-L2 {eh:.3e} vel {ev:.3e}; against the oracle on the shuffled input: h {eo:.3e}')
    assert eh <= 2e-6 and ev <= 1e-5 and eo <= P.FWD_TOLS['f16x3']
    assert float((out_p * (1 - inp_p['node_mask'].float())).abs().max()) == 0.0


@pytest.mark.parametrize('sizes,linkers', [([20, 12], [5, 4]), ([70, 30], [8, 4]), ([120, 12], [9, 4])])
def test_denoiser_without_context(sizes, linkers):
    """``context_node_nf = 0``: ``Dynamics.forward(..., context=None)`` appends nothing to the node features (egnn.py:403-407) - one
    compute unit per molecule, a team, the HBM-resident kernels."""
    from difflinker_amd import Dynamics
    nf, L = 8, 2
    dyn = Dynamics(n_dims=3, in_node_nf=nf, context_node_nf=0, hidden_nf=128, n_layers=L, norm_constant=1e-6)
    sd = seeded_state_dict(nf + 1, 128, L, 271)
    dyn.load_state_dict(sd, strict=True)
    dyn = dyn.to(P.dev())
    cfg = EGNNConfig(in_node_nf=nf, context_node_nf=0, n_layers=L)
    inp, z, t = P.ragged_inputs(sizes, linkers, nf, seed=272)
    ref = egnn_oracle.dynamics_forward(sd, cfg, t, z, inp['node_mask'], inp['linker_mask'], inp['edge_mask'], None)
    d = P.dev()
    out = dyn.forward(t.to(d), z.to(d), inp['node_mask'].to(d), inp['linker_mask'].to(d), inp['edge_mask'].to(d), None).cpu()
    ev, eh = P.report(f'no context, {sizes[0]} atoms', out, ref, z)
    assert ev <= P.FWD_TOLS['f16x3'] and eh <= P.FWD_TOLS['f16x3']


@pytest.mark.parametrize('norm_constant,normalization_factor', [(0.0, 100.0), (1.0, 100.0), (1e-6, 1.0), (0.5, 10.0)])
def test_other_norm_constants_and_normalization_factors(norm_constant, normalization_factor):
    """Every released configuration has ``norm_constant = 1e-6``, ``normalization_factor = 100``; the class defaults are 0 and 100
    (egnn.py:324-329) and both are free hyper-parameters: the coordinate difference is divided by ``norm + norm_constant``
    (egnn.py:240-247), message and coordinate sums by ``normalization_factor`` (egnn.py:294-301) - FC on one compute unit, a team
    and the HBM-resident kernels, and the radius-graph kernels."""
    from difflinker_amd import Dynamics, DynamicsWithPockets
    nf, L = 8, 2
    for sizes, linkers in (([20, 12], [5, 4]), ([70, 30], [8, 4]), ([120, 12], [9, 4])):
        dyn = Dynamics(n_dims=3, in_node_nf=nf, context_node_nf=1, hidden_nf=128, n_layers=L, norm_constant=norm_constant,
                       normalization_factor=normalization_factor)
        sd = seeded_state_dict(nf + 2, 128, L, 281, coord_gain=1.0 if normalization_factor == 100.0 else 0.02)
        dyn.load_state_dict(sd, strict=True)
        cfg = EGNNConfig(in_node_nf=nf, context_node_nf=1, n_layers=L, norm_constant=norm_constant, normalization_factor=normalization_factor)
        inp, z, t = P.ragged_inputs(sizes, linkers, nf, seed=282)
        ref = egnn_oracle.dynamics_forward(sd, cfg, t, z, inp['node_mask'], inp['linker_mask'], inp['edge_mask'], inp['context'])
        out = P.run_hip_forward(dyn.to(P.dev()), inp, z, t)
        ev, eh = P.report(f'norm_constant {norm_constant:g}, normalization_factor {normalization_factor:g}, {sizes[0]} atoms', out, ref, z)
        assert ev <= P.FWD_TOLS['f16x3'] and eh <= P.FWD_TOLS['f16x3']
    pdyn = DynamicsWithPockets(n_dims=3, in_node_nf=nf, context_node_nf=2, hidden_nf=128, n_layers=L, norm_constant=norm_constant,
                               normalization_factor=normalization_factor, graph_type='FC-10A-4A')
    sd = seeded_state_dict(nf + 3, 128, L, 283, coord_gain=0.02)
    pdyn.load_state_dict(sd, strict=True)
    cfg = EGNNConfig(in_node_nf=nf, context_node_nf=2, n_layers=L, norm_constant=norm_constant, normalization_factor=normalization_factor,
                     graph_type='FC-10A-4A')
    inp, z, t = P.pocket_inputs(batch=2, n_frag=12, n_pocket=60, linker=(4, 8), nf=nf, seed=284)
    ref = egnn_oracle.dynamics_forward_pockets(sd, cfg, t, z, inp['node_mask'], inp['linker_mask'], inp['edge_mask'], inp['context'])
    out = P.run_hip_forward(pdyn.to(P.dev()), inp, z, t)
    ev, eh = P.report(f'pockets, norm_constant {norm_constant:g}, normalization_factor {normalization_factor:g}', out, ref, z)
    assert ev <= P.FWD_TOLS['f16x3'] and eh <= P.FWD_TOLS['f16x3']


@pytest.mark.parametrize('kind', ['EDM', 'EDM on teams', 'InpaintingEDM'])
def test_sampler_with_other_schedule_and_normalisation(kind):
    """Every released configuration samples with polynomial_2 / 1e-5 / norm_values [1, 4, 10] / no bias; the sampler takes any
    ``polynomial_<p>`` schedule, precision, ``timesteps``, ``norm_values`` and ``norm_biases`` (edm.py:24-60, 347-361): here
    polynomial_3, 1e-4, a 200-step table sampled in 9 steps, x / 2, (h - 0.5) / 3."""
    from difflinker_amd import Dynamics, EDM, InpaintingEDM
    nf, L, T = 8, 2, 9
    inpaint = kind == 'InpaintingEDM'
    sizes, linkers = ([60, 20], [8, 4]) if kind == 'EDM on teams' else ([20, 12, 33], [5, 4, 7])
    dyn = Dynamics(n_dims=3, in_node_nf=nf, context_node_nf=1, hidden_nf=128, n_layers=L, norm_constant=1e-6, centering=inpaint)
    sd = seeded_state_dict(nf + 2, 128, L, 291)
    dyn.load_state_dict(sd, strict=True)
    cfg = EGNNConfig(in_node_nf=nf, context_node_nf=1, n_layers=L, centering=inpaint)
    kw = dict(in_node_nf=nf, timesteps=200, noise_schedule='polynomial_3', noise_precision=1e-4, norm_values=(2., 3., 5.), norm_biases=(None, 0.5, 0.))
    edm = (InpaintingEDM if inpaint else EDM)(dyn.to(P.dev()), n_dims=3, loss_type='l2', **kw).to(P.dev())
    edm.T = T
    inp, _, _ = P.ragged_inputs(sizes, linkers, nf, seed=292)
    B, N = inp['x'].shape[:2]
    bank = edm_oracle.NoiseBank.generate(2 * T + 1 if inpaint else T, B, N, 3, nf, seed=293)
    orc = (edm_oracle.InpaintingEDMOracle if inpaint else edm_oracle.EDMOracle)(edm_oracle.make_dynamics_oracle(sd, cfg), **kw)
    orc.T = T
    g = {k: v.to(P.dev()) for k, v in inp.items()}
    if inpaint:
        want = orc.sample_chain(inp['x'], inp['h'], inp['node_mask'], inp['edge_mask'], inp['fragment_mask'], inp['linker_mask'], inp['context'],
                                bank, keep_frames=3)
        got = edm.sample_chain(g['x'], g['h'], g['node_mask'], g['edge_mask'], g['fragment_mask'], g['linker_mask'], g['context'],
                               keep_frames=3, noise_bank=bank.stacked()).cpu()
        ex, efr = rel_l2(got[0, ..., :3], want[0, ..., :3]), rel_l2(got[1:], want[1:])
        print(f'[{kind}, polynomial_3 / 1e-4 / 200 steps / norm (2, 3) bias 0.5] x rel-L2 {ex:.3e} frames {efr:.3e}')
        assert ex <= P.CHAIN_TOL and efr <= P.CHAIN_TOL and torch.equal(got[0, ..., 3:], want[0, ..., 3:])
    else:
        want = orc.sample_chain(inp['x'], inp['h'], inp['node_mask'], inp['fragment_mask'], inp['linker_mask'], inp['edge_mask'], inp['context'],
                                bank, keep_frames=3)
        got = edm.sample_chain(g['x'], g['h'], g['node_mask'], g['fragment_mask'], g['linker_mask'], g['edge_mask'], g['context'],
                               keep_frames=3, noise_bank=bank.stacked()).cpu()
        P.check_chain(f'{kind}, polynomial_3 / 1e-4 / 200 steps / norm (2, 3) bias 0.5', got, want, inp)


@pytest.mark.parametrize('nf,ctx', [(10, 3), (13, 2), (1, 0), (4, 4)])
def test_other_feature_and_context_widths(nf, ctx):
    """``in_node_nf`` is the number of atom types (+ 1 with ``--include_charges``: train_difflinker.py:51-52), the context up to
    anchors + fragment + pocket masks: 8..10 and 1..3 in practice; the C ABI takes 3 + nf <= 16, context <= 4,
    nf + time + context <= 16.  Forward on one compute unit per molecule, a team and the HBM-resident kernels, then a short chain."""
    from difflinker_amd import Dynamics
    L, T = 2, 5
    dyn = Dynamics(n_dims=3, in_node_nf=nf, context_node_nf=ctx, hidden_nf=128, n_layers=L, norm_constant=1e-6)
    sd = seeded_state_dict(nf + ctx + 1, 128, L, 300 + nf, coord_gain=1.0)      # (a velocity of the order of the coordinates: above its ulp(|x|) floor)
    dyn.load_state_dict(sd, strict=True)
    dyn = dyn.to(P.dev())
    cfg = EGNNConfig(in_node_nf=nf, context_node_nf=ctx, n_layers=L)
    g = torch.Generator().manual_seed(301)
    d = P.dev()

    def with_context(inp):
        B, N = inp['x'].shape[:2]
        inp['context'] = (torch.randn(B, N, ctx, generator=g) * inp['node_mask'].float()) if ctx else None
        return inp
    for sizes, linkers in (([20, 12], [5, 4]), ([70, 30], [8, 4]), ([120, 12], [9, 4])):
        inp, z, t = P.ragged_inputs(sizes, linkers, nf, seed=302)
        inp = with_context(inp)
        ref = egnn_oracle.dynamics_forward(sd, cfg, t, z, inp['node_mask'], inp['linker_mask'], inp['edge_mask'], inp['context'])
        out = dyn.forward(t.to(d), z.to(d), inp['node_mask'].to(d), inp['linker_mask'].to(d), inp['edge_mask'].to(d),
                          inp['context'].to(d) if ctx else None).cpu()
        ev, eh = P.report(f'nf {nf}, context {ctx}, {sizes[0]} atoms', out, ref, z)
        assert ev <= P.FWD_TOLS['f16x3'] and eh <= P.FWD_TOLS['f16x3']
    inp, _, _ = P.ragged_inputs([20, 41, 60], [4, 7, 6], nf, seed=303)
    inp = with_context(inp)
    B, N = inp['x'].shape[:2]
    sd = seeded_state_dict(nf + ctx + 1, 128, L, 300 + nf)                      # (the chain: small steps)
    dyn.load_state_dict(sd, strict=True)
    dyn.invalidate_packed()
    edm = _edm(dyn, nf, T)
    bank = edm_oracle.NoiseBank.generate(T, B, N, 3, nf, seed=304)
    orc = edm_oracle.EDMOracle(edm_oracle.make_dynamics_oracle(sd, cfg), in_node_nf=nf, timesteps=500)
    orc.T = T
    want = orc.sample_chain(inp['x'], inp['h'], inp['node_mask'], inp['fragment_mask'], inp['linker_mask'], inp['edge_mask'], inp['context'],
                            bank, keep_frames=2)
    got = edm.sample_chain(inp['x'].to(d), inp['h'].to(d), inp['node_mask'].to(d), inp['fragment_mask'].to(d), inp['linker_mask'].to(d),
                           inp['edge_mask'].to(d), inp['context'].to(d) if ctx else None, keep_frames=2, noise_bank=bank.stacked()).cpu()
    P.check_chain(f'chain, nf {nf}, context {ctx}', got, want, inp)
