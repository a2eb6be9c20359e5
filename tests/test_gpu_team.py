"""Teams: several workgroups (compute units) per molecule on the LDS-resident path (``Dynamics.team``,
``dl_chain_args.team``, ``dl_egnn_forward_fc_team``).  A team splits the pair loop by receiving atom and exchanges the message
sums through HBM once per pass; every team size must give the oracle's numbers (same tolerances as one workgroup per
molecule), bitwise repeatably, for ragged sizes down to fewer atoms than team members, for batches that are not a multiple
of 8 (team members sit 8 workgroups apart) and under uneven load inside a team's group of 8 molecules.

The rest of the GPU suite runs with ``team='auto'`` (its batches are small, so it exercises teams of 4 throughout); the
cases here pin the team size, 1 included."""
import ctypes

import pytest
import torch

from helpers import rel_l2
from oracle import edm_oracle, egnn_oracle
from test_gpu_parity import (CHAIN_TOL, FWD_TOLS, chain_case, check_chain, dev, make_dynamics, ragged_inputs, report, run_hip_forward)

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('team', [1, 2, 4, 8])
@pytest.mark.parametrize('precision', ['f16x3', 'fp32'])
def test_forward_vs_oracle_per_team_size(team, precision):
    nf, L = 9, 2
    # 55 = LDS limit; 3 and 2 atoms: fewer receivers than team members (some members own no atom); 13 molecules: not a multiple of 8
    sizes = [55, 3, 40, 2, 33, 2, 50, 9, 32, 31, 17, 5, 48]
    linkers = [6, 1, 12, 1, 5, 1, 8, 3, 4, 4, 6, 2, 10]
    dyn, sd, cfg = make_dynamics(nf, 1, L, seed=300 + team, precision=precision)
    dyn.team = team
    inp, z, t = ragged_inputs(sizes, linkers, nf, seed=77)
    ref = egnn_oracle.dynamics_forward(sd, cfg, t, z, inp['node_mask'], inp['linker_mask'], inp['edge_mask'], inp['context'])
    out = run_hip_forward(dyn, inp, z, t)
    ev, eh = report(f'team {team} {precision}', out, ref, z)
    assert ev <= FWD_TOLS[precision] and eh <= FWD_TOLS[precision]
    assert torch.equal(out, run_hip_forward(dyn, inp, z, t)), 'bitwise repeatable for a given team size'
    assert float(out[inp['node_mask'].expand_as(out) == 0].abs().max()) == 0.0


def test_team_sizes_agree_to_rounding_and_full_depth():
    """GEOM depth (6 blocks = 18 exchanges per forward), every team size against team = 1 and the oracle."""
    nf, L = 9, 6
    sizes, linkers = [50, 35, 44, 47, 38, 41, 36, 49, 50, 42], [8, 3, 12, 9, 5, 7, 4, 11, 10, 6]
    dyn, sd, cfg = make_dynamics(nf, 1, L, seed=321)
    inp, z, t = ragged_inputs(sizes, linkers, nf, seed=78)
    ref = egnn_oracle.dynamics_forward(sd, cfg, t, z, inp['node_mask'], inp['linker_mask'], inp['edge_mask'], inp['context'])
    outs = {}
    for team in (1, 2, 4, 8):
        dyn.team = team
        outs[team] = run_hip_forward(dyn, inp, z, t)
        ev, eh = report(f'6 blocks, team {team}', outs[team], ref, z)
        assert ev <= FWD_TOLS['f16x3'] and eh <= FWD_TOLS['f16x3']
    for team in (2, 4, 8):
        assert rel_l2(outs[team][..., 3:], outs[1][..., 3:]) <= 2e-6


@pytest.mark.parametrize('team', [1, 2, 4, 8])
def test_chain_vs_oracle_per_team_size(team):
    from difflinker_amd import EDM
    nf, L, T, keep = 8, 2, 12, 3
    sizes, linkers = [12, 7, 10, 30, 2, 21, 16, 9, 26], [4, 2, 3, 9, 1, 6, 5, 3, 8]
    dyn, sd, cfg = make_dynamics(nf, 1, L, seed=40)
    dyn.team = team
    inp, _, _ = ragged_inputs(sizes, linkers, nf, seed=41)
    B, N = inp['x'].shape[:2]
    edm = EDM(dyn, in_node_nf=nf, n_dims=3, timesteps=500, noise_schedule='polynomial_2', noise_precision=1e-5,
              loss_type='l2', norm_values=[1, 4, 10]).to(dev())
    edm.T = T
    bank = edm_oracle.NoiseBank.generate(T, B, N, 3, nf, seed=42)
    orc = edm_oracle.EDMOracle(edm_oracle.make_dynamics_oracle(sd, cfg), in_node_nf=nf, timesteps=500)
    orc.T = T
    want = orc.sample_chain(inp['x'], inp['h'], inp['node_mask'], inp['fragment_mask'], inp['linker_mask'],
                            inp['edge_mask'], inp['context'], bank, keep_frames=keep)
    g = {k: v.to(dev()) for k, v in inp.items()}
    run = lambda: edm.sample_chain(g['x'], g['h'], g['node_mask'], g['fragment_mask'], g['linker_mask'],    # noqa: E731
                                   g['edge_mask'], g['context'], keep_frames=keep, noise_bank=bank.stacked()).cpu()
    got = run()
    check_chain(f'chain T=12, team {team}', got, want, inp)
    assert torch.equal(got, run()), 'bitwise repeatable for a given team size'


@pytest.mark.parametrize('batch,teams', [(64, (4, 2)), (24, (8,))])
def test_full_length_chain_teams_against_single_workgroup(batch, teams):
    """The reference's default sampling batch (64 molecules, generate.py:145) - and a smaller one, for teams of 8 - at GEOM
    size and depth, T = 500: 9018 exchanges per team.  A stale or torn exchange row anywhere would show as a gross error
    against team = 1."""
    from difflinker_amd import EDM, synthetic
    nf, L = 9, 6
    dyn, sd, cfg = make_dynamics(nf, 1, L, seed=81, coord_gain=0.001)
    data, _ = synthetic.make_batch('C2', seed=5, batch=batch)
    inp = {k: v.to(dev()) for k, v in synthetic.sampler_inputs(data).items()}
    edm = EDM(dyn, in_node_nf=nf, n_dims=3, timesteps=500, noise_schedule='polynomial_2', noise_precision=1e-5,
              loss_type='l2', norm_values=[1, 4, 10]).to(dev())
    edm.noise_source = 'philox'
    chains = {}
    for team in (1,) + tuple(teams):
        dyn.team = team
        edm.noise_seed = 11
        chains[team] = edm.sample_chain(inp['x'], inp['h'], inp['node_mask'], inp['fragment_mask'], inp['linker_mask'],
                                        inp['edge_mask'], inp['context'], keep_frames=1)[0].cpu()
        assert torch.isfinite(chains[team]).all()
    lm = inp['linker_mask'].cpu()
    for team in teams:
        ex = rel_l2(chains[team][..., :3] * lm, chains[1][..., :3] * lm)
        mism = int((chains[team][..., 3:] != chains[1][..., 3:]).any(-1).sum())
        print(f'[T=500, B={batch}, team {team} vs 1] linker-x rel-L2 {ex:.3e}, one-hot mismatches {mism}')
        assert ex <= CHAIN_TOL and mism == 0
    dyn.team = 'auto'
    assert dyn.team_for(32) == 8 and dyn.team_for(64) == 4 and dyn.team_for(128) == 2 and dyn.team_for(129) == 1 and dyn.team_for(4096) == 1


def test_team_requests_the_device_cannot_hold_are_refused():
    from difflinker_amd import _lib
    lib = _lib.load()
    assert lib.dl_team_max(8) == 8 and lib.dl_team_max(32) == 8 and lib.dl_team_max(33) == 4 and lib.dl_team_max(64) == 4
    assert lib.dl_team_max(65) == 2 and lib.dl_team_max(128) == 2 and lib.dl_team_max(256) == 1
    assert lib.dl_workspace_bytes(0, 4) == 0 and lib.dl_workspace_bytes(3, 4) > lib.dl_workspace_bytes(3, 2) > lib.dl_workspace_bytes(3, 1) > 0
    nf = 9
    dyn, sd, cfg = make_dynamics(nf, 1, 1, seed=5)
    inp, z, t = ragged_inputs([10] * 70, [3] * 70, nf, seed=6)          # 70 molecules: 72 slots x 4 > 256 compute units
    for too_many in (4, 8):
        dyn.team = too_many
        with pytest.raises(_lib.HipLibraryError):
            run_hip_forward(dyn, inp, z, t)
    dyn.team = 2
    ref = egnn_oracle.dynamics_forward(sd, cfg, t, z, inp['node_mask'], inp['linker_mask'], inp['edge_mask'], inp['context'])
    ev, eh = report('70 molecules, team 2', run_hip_forward(dyn, inp, z, t), ref, z)
    assert ev <= FWD_TOLS['f16x3'] and eh <= FWD_TOLS['f16x3']
    dyn.team = 3
    with pytest.raises(ValueError):
        run_hip_forward(dyn, inp, z, t)


def test_a_team_that_cannot_assemble_fails_together_and_the_call_is_rerun_without_teams():
    """ADVICE round 2 / VERDICT item 4.  `dl_debug_team_fault(n)` makes member 1 of every team of the next n team launches give up at its first exchange, as a
    member whose team-mates never became resident would after its spin limit: it publishes the poison arrival word, every
    other member stops waiting too, ALL of them end with flag bit 3 (raw C ABI), and the Python drop-in re-runs the batch on
    one compute unit per molecule - the caller sees the oracle's numbers and no exception."""
    from difflinker_amd import _lib
    assert not hasattr(_lib.load(), 'dl_debug_team_fault'), 'the product library must not export the test hook'
    nf = 9
    dyn, sd, cfg = make_dynamics(nf, 1, 2, seed=31)
    inp, z, t = ragged_inputs([20, 33, 9, 50], [4, 6, 2, 9], nf, seed=17)
    ref = egnn_oracle.dynamics_forward(sd, cfg, t, z, inp['node_mask'], inp['linker_mask'], inp['edge_mask'], inp['context'])
    dyn.team = 4
    # the hook lives in the -DDL_TEST_HOOKS build of the library only (libdifflinker_hip_testhooks.so): inside this context the
    # package's launches go through that build
    with _lib.test_hooks() as lib:
        lib.dl_debug_team_fault(3)          # the three team launches below fail (the count runs down by itself)
        prep = dyn.prepare(inp['node_mask'].to(dev()), inp['linker_mask'].to(dev()), inp['edge_mask'].to(dev()), inp['context'].to(dev()))
        out, flags = dyn.launch(prep, t.to(dev()), z.to(dev()))
        torch.cuda.synchronize()
        assert all(int(f) & 8 for f in flags.cpu().tolist()), 'every molecule of a failed launch is flagged void'
        ev, eh = report('team fault -> rerun on one compute unit', run_hip_forward(dyn, inp, z, t), ref, z)
        assert ev <= FWD_TOLS['f16x3'] and eh <= FWD_TOLS['f16x3']
        # the fused chain too (same draws in both runs: the bank is fixed before the first launch)
        got, want, cinp = chain_case(nf=9, n_layers=2, sizes=[22, 35], linkers=[5, 7], T=20, keep=1, seed=33, team=2)
        check_chain('chain, team fault -> rerun', got, want, cinp)


@pytest.mark.parametrize('precision', ['f16x3', 'fp32'])
@pytest.mark.parametrize('sizes,linkers', [([56, 10], [5, 2]), ([70, 33, 110, 64], [9, 4, 12, 11])])
def test_molecules_of_56_to_110_atoms_run_fused_on_teams(sizes, linkers, precision):
    """a molecule beyond one compute unit's LDS takes a team of at least two (each member holds its own
    atoms' state and every atom's sender row), no host-driven loop: forward and chain against the oracle."""
    from difflinker_amd import _lib
    nf = 9
    assert _lib.load().dl_team_max_atoms(2) >= 110 and _lib.load().dl_team_max_atoms(1) == _lib.load().dl_max_atoms() == 55
    dyn, sd, cfg = make_dynamics(nf, 1, 2, seed=41, precision=precision)
    inp, z, t = ragged_inputs(sizes, linkers, nf, seed=sum(sizes))
    ref = egnn_oracle.dynamics_forward(sd, cfg, t, z, inp['node_mask'], inp['linker_mask'], inp['edge_mask'], inp['context'])
    for team in ('auto', 2):
        dyn.team = team
        prep = dyn.prepare(inp['node_mask'].to(dev()), inp['linker_mask'].to(dev()), inp['edge_mask'].to(dev()), inp['context'].to(dev()))
        kinds = [(p_['large'], p_['team']) for p_ in prep.get('split', [prep])]
        assert not any(large for large, _ in kinds), 'no molecule of this batch needs the HBM-resident kernels'
        ev, eh = report(f'fwd sizes={sizes} team={team} {precision}', run_hip_forward(dyn, inp, z, t), ref, z)
        assert ev <= FWD_TOLS[precision] and eh <= FWD_TOLS[precision]
    got, want, cinp = chain_case(nf=9, n_layers=2, sizes=sizes, linkers=linkers, T=30, keep=3, seed=43, precision=precision)
    check_chain(f'chain sizes={sizes} {precision}', got, want, cinp)
