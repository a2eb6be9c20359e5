"""CPU tests of the host side: C-ABI library exports, parameter containers / checkpoint keys,
per-step scalar tables, DDPM glue, batch sharding (gloo, world_size 2)."""
import ctypes
import os
import re
import sys

import pytest
import torch
import torch.multiprocessing as mp

from helpers import dynamics_param_shapes, seeded_state_dict, rel_l2
from oracle import edm_oracle, egnn_oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_loads_and_exports_every_declared_symbol():
    import __graft_entry__ as g
    g.build()
    from difflinker_amd import _lib
    lib = _lib.load()
    header = open(os.path.join(ROOT, 'include', 'difflinker_hip.h')).read()
    hooks = re.findall(r'#ifdef DL_TEST_HOOKS(.*?)#endif', header, re.S)
    declared = set(re.findall(r'\b(dl_[a-z_0-9]+)\s*\(', re.sub(r'#ifdef DL_TEST_HOOKS.*?#endif', '', header, flags=re.S)))
    hooked = set(re.findall(r'\b(dl_[a-z_0-9]+)\s*\(', ' '.join(hooks)))
    assert declared == set(_lib.EXPORTS), (declared ^ set(_lib.EXPORTS))
    assert hooked == set(_lib.TEST_HOOK_EXPORTS), (hooked ^ set(_lib.TEST_HOOK_EXPORTS))
    for name in declared:
        assert hasattr(lib, name), name
    for name in hooked:                      # test hooks: in the -DDL_TEST_HOOKS build only, never in the product library
        assert not hasattr(lib, name), f'{name} exported by the product library'
        assert hasattr(ctypes.CDLL(_lib.TEST_HOOKS_LIB_PATH), name), name
    assert lib.dl_abi_version() == _lib.ABI_VERSION == 7
    # the caller-owned scratch (ABI v6): a size query, no device needed; nothing for an empty batch, linear in the batch,
    # a team adds its exchange rows and arrival words on top of one h-row block per workgroup
    w1, w2 = lib.dl_workspace_bytes(1, 1), lib.dl_workspace_bytes(2, 1)
    assert lib.dl_workspace_bytes(0, 1) == 0 and w1 > 0 and w2 == 2 * w1 and w1 % 16 == 0
    assert lib.dl_workspace_bytes(8, 0) == lib.dl_workspace_bytes(8, 1) == 8 * w1
    assert lib.dl_workspace_bytes(8, 4) > 4 * lib.dl_workspace_bytes(8, 1) and lib.dl_workspace_bytes(8, 3) >= 0
    assert lib.dl_team_max(64) in (1, 2, 4, 8) and lib.dl_team_max(0) == 1         # 1 without a device (a query, not a compute call)
    assert ctypes.sizeof(_lib.DLChainArgs) == 232                                   # dl_chain_args of ABI v7 (LP64): v6's 192 + order_first, order_count + q_begin, q_end, z_state, skip_flags
    assert lib.dl_max_atoms() == 55
    assert lib.dl_error_string(-2).decode().startswith('hyper-parameter')
    cfg = _lib.DLConfig(3, 9, 1, 128, 6, 2, 1, 1e-6, 100.0, 1)
    assert lib.dl_model_num_tensors(ctypes.byref(cfg)) == 4 + 6 * 21
    scfg = _lib.DLSizeConfig(8, 128, 10, 3)
    assert lib.dl_size_model_num_tensors(ctypes.byref(scfg)) == 4 + 8 * 3
    assert lib.dl_size_model_num_tensors(ctypes.byref(_lib.DLSizeConfig(8, 64, 10, 3))) == -2
    assert lib.dl_size_max_fragment_atoms() == 64


def test_no_gpu_means_loud_failure_not_fallback():
    from difflinker_amd import Dynamics
    dyn = Dynamics(3, 9, 1, hidden_nf=128, n_layers=1, norm_constant=1e-6)
    xh = torch.zeros(1, 4, 12)
    with pytest.raises(RuntimeError, match='GPU only'):
        dyn.forward(torch.zeros(1, 1), xh, torch.ones(1, 4, 1, dtype=torch.int8), torch.ones(1, 4, 1),
                    torch.ones(16, 1, dtype=torch.int8), torch.ones(1, 4, 1))


def test_team_knob_of_the_dynamics():
    """``Dynamics.team``: 'auto' asks the library (1 without a device), explicit sizes are 1, 2, 4 or 8, anything else raises."""
    from difflinker_amd import Dynamics
    dyn = Dynamics(3, 9, 1, hidden_nf=128, n_layers=1, norm_constant=1e-6)
    assert dyn.team == 'auto' and dyn.team_for(64) in (1, 2, 4, 8)
    for team in (1, 2, 4, 8, '4'):
        dyn.team = team
        assert dyn.team_for(8) == int(team)
    for bad in (3, 0, 16):
        dyn.team = bad
        with pytest.raises(ValueError):
            dyn.team_for(8)


def test_unsupported_hparams_raise():
    from difflinker_amd import Dynamics
    for kw in (dict(aggregation_method='max'), dict(hidden_nf=256), dict(inv_sublayers=5), dict(model='gnn_dynamics')):
        args = dict(n_dims=3, in_node_nf=9, context_node_nf=1, hidden_nf=128, n_layers=1)
        args.update(kw)
        with pytest.raises(NotImplementedError):
            Dynamics(**args)


def test_narrow_network_padded_to_the_kernel_width_is_the_same_function():
    """hidden_nf <= 128 (the reference's default is 64, egnn.py:324-329) runs on the 128-wide kernels with zero-padded weights
    (``egnn.pad_to_kernel_width``): the padded state_dict, read as a 128-wide network, must compute what the narrow one does -
    checked here on the oracle, which takes any width (the GPU tests then hold the kernels to the NARROW oracle)."""
    from difflinker_amd import Dynamics, synthetic
    from difflinker_amd.egnn import egnn_tensor_order, pad_to_kernel_width
    nf, L, sub, h = 8, 2, 3, 64
    dyn = Dynamics(n_dims=3, in_node_nf=nf, context_node_nf=1, hidden_nf=h, n_layers=L, inv_sublayers=sub, attention=True,
                   condition_time=False, norm_constant=1e-6)
    sd = seeded_state_dict(nf + 1, h, L, 21, inv_sublayers=sub, attention=True)
    dyn.load_state_dict(sd, strict=True)
    cfg = dyn.hip_config()
    assert (cfg.hidden_nf, cfg.inv_sublayers, cfg.condition_time) == (128, sub, 0)
    wide = {'dynamics.' + k: pad_to_kernel_width(k, dyn.dynamics.state_dict()[k], h)
            for k in egnn_tensor_order(L, inv_sublayers=sub, attention=True)}
    assert wide['dynamics.e_block_1.gcl_2.edge_mlp.0.weight'].shape == (128, 258)
    data, _ = synthetic.make_batch('C1', seed=5, batch=3)
    inp = synthetic.sampler_inputs(data)
    B, N = inp['x'].shape[:2]
    g = torch.Generator().manual_seed(2)
    z = torch.cat([inp['x'], inp['h']], dim=2) * inp['fragment_mask'] + torch.randn((B, N, 3 + nf), generator=g) * inp['linker_mask']
    t = torch.full((B, 1), 0.3)
    kw = dict(in_node_nf=nf, context_node_nf=1, n_layers=L, inv_sublayers=sub, attention=True, condition_time=False)
    narrow = egnn_oracle.dynamics_forward(sd, egnn_oracle.EGNNConfig(hidden_nf=h, **kw), t, z, inp['node_mask'], inp['linker_mask'],
                                          inp['edge_mask'], inp['context'])
    padded = egnn_oracle.dynamics_forward(wide, egnn_oracle.EGNNConfig(hidden_nf=128, **kw), t, z, inp['node_mask'],
                                          inp['linker_mask'], inp['edge_mask'], inp['context'])
    assert rel_l2(padded, narrow) <= 1e-6


def test_optional_hparams_own_the_reference_parameters():
    """attention / tanh / mean are accepted on the fully-connected path; attention adds ``att_mlp.0`` to every GCL in the
    reference's registration order (egnn.py:42-43), and the C-ABI tensor order carries it."""
    from difflinker_amd import Dynamics
    from difflinker_amd.egnn import egnn_tensor_order
    dyn = Dynamics(3, 9, 1, hidden_nf=128, n_layers=2, norm_constant=1e-6, attention=True, tanh=True, aggregation_method='mean')
    expect = dynamics_param_shapes(11, 128, 2, attention=True)
    assert [k for k, _ in dyn.state_dict().items()] == [e[0] for e in expect]
    order = egnn_tensor_order(2, attention=True)
    assert len(order) == 4 + 2 * (2 * 10 + 5) and 'e_block_1.gcl_0.att_mlp.0.bias' in order
    cfg = dyn.hip_config()
    assert (cfg.attention, cfg.tanh, cfg.aggregation_mean, cfg.sin_embedding) == (1, 1, 1, 0) and cfg.coords_range == 15.0


def test_sin_embedding_widens_the_edge_mlps():
    """sin_embedding=True: 24 edge attributes instead of 2 (egnn.py:193-198), flagged in the C-ABI config."""
    from difflinker_amd import Dynamics
    dyn = Dynamics(3, 9, 1, hidden_nf=128, n_layers=2, norm_constant=1e-6, sin_embedding=True)
    expect = dynamics_param_shapes(11, 128, 2, edge_feat_nf=24)
    assert [(k, tuple(v.shape)) for k, v in dyn.state_dict().items()] == [(e[0], tuple(e[1])) for e in expect]
    assert dyn.state_dict()['dynamics.e_block_0.gcl_equiv.coord_mlp.0.weight'].shape == (128, 280)
    assert dyn.hip_config().sin_embedding == 1


def test_state_dict_keys_and_tensor_order():
    from difflinker_amd import Dynamics
    from difflinker_amd.egnn import egnn_tensor_order
    dyn = Dynamics(3, 9, 1, hidden_nf=128, n_layers=3, norm_constant=1e-6)
    expect = dynamics_param_shapes(11, 128, 3)
    sd = dyn.state_dict()
    assert list(sd.keys()) == [k for k, *_ in expect]
    for k, shape, *_ in expect:
        assert tuple(sd[k].shape) == shape
    order = egnn_tensor_order(3)
    assert sorted('dynamics.' + k for k in order) == sorted(sd.keys())
    dyn.load_state_dict(seeded_state_dict(11, 128, 3, seed=1), strict=True)


@pytest.mark.skipif(not os.path.isdir('/root/reference/src'), reason='reference tree only exists in the build container')
def test_same_seed_same_init_as_reference():
    sys.path.insert(0, '/root/reference')
    sys.dont_write_bytecode = True
    from src.egnn import Dynamics as RefDynamics
    from difflinker_amd import Dynamics
    kw = dict(n_dims=3, in_node_nf=8, context_node_nf=1, hidden_nf=128, n_layers=2, norm_constant=1e-6)
    torch.manual_seed(7)
    ref = RefDynamics(**kw)
    torch.manual_seed(7)
    ours = Dynamics(**kw)
    a, b = ref.state_dict(), ours.state_dict()
    assert list(a.keys()) == list(b.keys())
    for k in a:
        assert torch.equal(a[k], b[k]), k


@pytest.mark.parametrize('timesteps,schedule,precision', [(500, 'polynomial_2', 1e-5), (1000, 'polynomial_2', 1e-5), (200, 'polynomial_3', 1e-4),
                                                          (100, 'polynomial_1', 1e-3)])
def test_step_coefficients_match_oracle_scalars(timesteps, schedule, precision):
    from difflinker_amd import Dynamics, EDM
    dyn = Dynamics(3, 8, 1, hidden_nf=128, n_layers=1, norm_constant=1e-6)
    for T in (timesteps, 50, 12):
        edm = EDM(dyn, in_node_nf=8, n_dims=3, timesteps=timesteps, noise_schedule=schedule, noise_precision=precision,
                  loss_type='l2', norm_values=[1, 4, 10])
        edm.T = T
        coefs, (inv_a0, s0, sx) = edm.step_coefficients()
        orc = edm_oracle.EDMOracle(None, in_node_nf=8, timesteps=timesteps, noise_schedule=schedule, noise_precision=precision)
        orc.T = T
        assert torch.equal(edm.gamma.gamma.data, orc.gamma_table)
        z = torch.zeros(1, 1, 1)
        for q, s in enumerate(reversed(range(T))):
            s_arr = torch.full((1, 1), s) / T
            t_arr = (torch.full((1, 1), s) + 1) / T
            g_s, g_t = orc.gamma(s_arr), orc.gamma(t_arr)
            s2, s_ts, a_ts = orc.sigma_and_alpha_t_given_s(g_t, g_s, z)
            want = torch.stack([t_arr.view(()), a_ts.view(()), (s2 / a_ts / orc.sigma(g_t, z)).view(()),
                                (s_ts * orc.sigma(g_s, z) / orc.sigma(g_t, z)).view(())])
            assert torch.equal(coefs[q], want), (T, q)
        g0 = orc.gamma(torch.zeros(1, 1))
        assert abs(sx - float(torch.exp(0.5 * g0))) < 1e-9
        assert abs(inv_a0 - float(1. / orc.alpha(g0, z))) < 1e-7 and abs(s0 - float(orc.sigma(g0, z))) < 1e-9


def test_ddpm_checkpoint_roundtrip(tmp_path):
    from difflinker_amd import DDPM
    hp = dict(in_node_nf=8, n_dims=3, context_node_nf=1, hidden_nf=128, activation='silu', tanh=False, n_layers=2,
              attention=False, norm_constant=1e-6, inv_sublayers=2, sin_embedding=False, normalization_factor=100,
              aggregation_method='sum', diffusion_steps=500, diffusion_noise_schedule='polynomial_2',
              diffusion_noise_precision=1e-5, diffusion_loss_type='l2', normalize_factors=[1, 4, 10],
              include_charges=False, model='egnn_dynamics', data_path='d', train_data_prefix='zinc_final_train',
              val_data_prefix='zinc_final_val', batch_size=8, lr=2e-4, torch_device='cpu', test_epochs=20,
              n_stability_samples=10, normalization='batch_norm', anchors_context=False)
    torch.manual_seed(3)
    m = DDPM(**hp)
    keys = list(m.state_dict().keys())
    assert keys[0] == 'edm.gamma.gamma' and 'edm.dynamics.dynamics.embedding.weight' in keys
    assert 'edm.dynamics.dynamics.e_block_1.gcl_equiv.coord_mlp.4.weight' in keys
    path = tmp_path / 'm.ckpt'
    torch.save(m.checkpoint_dict(), path)
    m2 = DDPM.load_from_checkpoint(str(path), map_location='cpu')
    for k, v in m.state_dict().items():
        assert torch.equal(v, m2.state_dict()[k])
    assert m2.edm.T == 500 and not m2.is_geom and m2.center_of_mass == 'fragments'


def _gloo_worker(rank, world, port, tmp):
    import torch.distributed as dist
    from difflinker_amd import synthetic
    from difflinker_amd.distributed import sample_chain_sharded, shard_bounds
    dist.init_process_group('gloo', init_method=f'tcp://127.0.0.1:{port}', rank=rank, world_size=world)
    torch.set_num_threads(2)
    data, cfg = synthetic.make_batch('C1', seed=9, batch=5)
    inp = synthetic.sampler_inputs(data)
    nf, L, T = cfg['nf'], 1, 3
    sd = seeded_state_dict(nf + 2, 128, L, seed=4)
    ocfg = egnn_oracle.EGNNConfig(in_node_nf=nf, context_node_nf=1, n_layers=L)
    B, N = inp['x'].shape[:2]
    bank = edm_oracle.NoiseBank.generate(T, B, N, 3, nf, seed=6)

    class OracleEDM:
        """Stand-in with the product EDM's sample_chain signature (the HIP EDM needs a GPU)."""
        coef_batch = None
        team_batch = None

        def sample_chain(self, x, h, node_mask, fragment_mask, linker_mask, edge_mask, context, keep_frames=None,
                         noise_bank=None):
            # a shard samples with the per-step scalars and the team size of the WHOLE batch (EDM.coef_batch / team_batch)
            assert self.coef_batch == 5 and self.team_batch == 5 and x.shape[0] in (2, 3)
            o = edm_oracle.EDMOracle(edm_oracle.make_dynamics_oracle(sd, ocfg), in_node_nf=nf, timesteps=500)
            o.T = T
            draws = []
            for k in range(T + 2):
                draws += [noise_bank[0][k], noise_bank[1][k]]
            return o.sample_chain(x, h, node_mask, fragment_mask, linker_mask, edge_mask, context,
                                  edm_oracle.NoiseBank(draws), keep_frames=keep_frames)

    stand_in = OracleEDM()
    full = sample_chain_sharded(stand_in, inp, keep_frames=2, noise_bank=bank.stacked())
    assert stand_in.coef_batch is None and stand_in.team_batch is None       # pins released after the call
    lo, hi = shard_bounds(B, rank, world)
    assert (hi - lo) in (2, 3)
    if rank == 0:
        torch.save(full, os.path.join(tmp, 'sharded.pt'))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_chain_equals_single_process_gloo(tmp_path):
    """world_size-2 run (gloo, CPU) of the sharding + all-gather path == unsharded oracle chain."""
    from difflinker_amd import synthetic
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_gloo_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    got = torch.load(os.path.join(str(tmp_path), 'sharded.pt'))
    data, cfg = synthetic.make_batch('C1', seed=9, batch=5)
    inp = synthetic.sampler_inputs(data)
    nf, L, T = cfg['nf'], 1, 3
    sd = seeded_state_dict(nf + 2, 128, L, seed=4)
    ocfg = egnn_oracle.EGNNConfig(in_node_nf=nf, context_node_nf=1, n_layers=L)
    B, N = inp['x'].shape[:2]
    bank = edm_oracle.NoiseBank.generate(T, B, N, 3, nf, seed=6)
    o = edm_oracle.EDMOracle(edm_oracle.make_dynamics_oracle(sd, ocfg), in_node_nf=nf, timesteps=500)
    o.T = T
    want = o.sample_chain(inp['x'], inp['h'], inp['node_mask'], inp['fragment_mask'], inp['linker_mask'],
                          inp['edge_mask'], inp['context'], bank, keep_frames=2)
    assert got.shape == want.shape
    assert torch.equal(got, want)      # per-molecule arithmetic is batch-independent on the oracle


def test_shard_bounds_cover_batch():
    from difflinker_amd.distributed import shard_bounds
    for n in (1, 5, 8, 256, 2048):
        for w in (1, 2, 3, 4, 8):
            spans = [shard_bounds(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


@pytest.mark.parametrize('norm', [None, 'batch_norm'])
def test_size_gnn_state_dict_and_loud_failure(norm):
    """SizeGNN / SizeClassifier hold the reference's parameters under the reference's keys (linker_size.py:45-81,
    linker_size_lightning.py:45-52); CPU tensors raise instead of falling back."""
    from helpers import size_gnn_param_shapes, seeded_size_state_dict
    from difflinker_amd.linker_size import SizeClassifier, SizeGNN
    from difflinker_amd import _lib
    gnn = SizeGNN(in_node_nf=8, hidden_nf=128, out_node_nf=10, n_layers=3, normalization=norm)
    want = [(k, tuple(shape)) for k, shape, _, _ in size_gnn_param_shapes(8, 128, 10, 3, batch_norm=norm is not None)]
    got = [(k, tuple(v.shape)) for k, v in gnn.state_dict().items()]
    assert got == want
    clf = SizeClassifier(in_node_nf=8, hidden_nf=128, out_node_nf=10, n_layers=3, normalization=norm).eval()
    clf.load_state_dict(seeded_size_state_dict(8, 128, 10, 3, seed=1, batch_norm=norm is not None, prefix='gnn.'), strict=True)
    assert clf.linker_id2size[0] == 3 and clf.linker_size2id[12] == 9
    data = {'one_hot': torch.zeros(1, 4, 8), 'positions': torch.zeros(1, 4, 3), 'fragment_mask': torch.ones(1, 4, 1),
            'edge_mask': torch.ones(16, 1)}
    with pytest.raises(_lib.HipLibraryError, match='no CPU fallback'):
        clf.forward(data, return_loss=False)
    with pytest.raises(NotImplementedError):
        SizeGNN(8, 256, 10, 3, None)


def test_collate_with_fragment_edges_conventions():
    """Fragment-only edge mask with the int8 ``~eye`` values (-1 off / -2 on the diagonal) and the FC edge list
    (datasets.py:378-422)."""
    from difflinker_amd.datasets import collate_with_fragment_edges
    mols = []
    for n, nl in [(3, 1), (2, 0)]:
        frag = torch.zeros(n)
        frag[:n - nl] = 1
        mols.append({'positions': torch.randn(n, 3), 'one_hot': torch.eye(8)[:n], 'anchors': torch.zeros(n),
                     'fragment_mask': frag, 'linker_mask': 1 - frag, 'num_atoms': n, 'uuid': 0, 'name': 'm'})
    out = collate_with_fragment_edges(mols)
    em = out['edge_mask'].view(2, 3, 3)
    assert em.dtype == torch.float32
    assert em[0].tolist() == [[-2, -1, 0], [-1, -2, 0], [0, 0, 0]]
    assert em[1].tolist() == [[-2, -1, 0], [-1, -2, 0], [0, 0, 0]]
    rows, cols = out['edges']
    assert rows.tolist()[:9] == [0, 0, 0, 1, 1, 1, 2, 2, 2] and cols.tolist()[:9] == [0, 1, 2] * 3
    assert rows.tolist()[9:12] == [3, 3, 3] and cols.tolist()[9:12] == [3, 4, 5]
    assert out['atom_mask'].shape == (2, 3, 1) and out['fragment_mask'].shape == (2, 3, 1)


def _toy_dataset(n_mols, nf, pockets=False, seed=0):
    g = torch.Generator().manual_seed(seed)
    data = []
    for k in range(n_mols):
        n_frag, n_link, n_pock = 6 + k, 3, (5 if pockets else 0)
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
        data.append(item)
    return data


def test_datasets_setup_and_resume_logic(tmp_path):
    """Preprocessed-dataset loaders (datasets.py:40-54, :103-129), DDPM.setup('val') / val_dataloader
    (lightning.py:115-146) and sample.py's resume rule (sample.py:37-60)."""
    from difflinker_amd import DDPM
    from difflinker_amd.datasets import MOADDataset, ZincDataset, collate_with_fragment_edges
    from difflinker_amd.sample import check_if_generated
    torch.save(_toy_dataset(5, 8), os.path.join(tmp_path, 'zinc_final_test.pt'))
    torch.save(_toy_dataset(3, 9, pockets=True), os.path.join(tmp_path, 'MOAD_test_full.pt'))
    assert len(ZincDataset(str(tmp_path), 'zinc_final_test', 'cpu')) == 5
    assert len(MOADDataset(data_path=str(tmp_path), prefix='MOAD_test.full', device='cpu')) == 3
    assert len(MOADDataset(data_path=str(tmp_path), prefix='MOAD_test_full', device='cpu')) == 3
    with pytest.raises(FileNotFoundError, match='preprocess'):
        ZincDataset(str(tmp_path), 'zinc_final_train', 'cpu')
    hp = dict(in_node_nf=8, n_dims=3, context_node_nf=1, hidden_nf=128, activation='silu', tanh=False, n_layers=1,
              attention=False, norm_constant=1e-6, inv_sublayers=2, sin_embedding=False, normalization_factor=100,
              aggregation_method='sum', diffusion_steps=500, diffusion_noise_schedule='polynomial_2',
              diffusion_noise_precision=1e-5, diffusion_loss_type='l2', normalize_factors=[1, 4, 10],
              include_charges=False, model='egnn_dynamics', data_path=str(tmp_path), train_data_prefix='zinc_final_train',
              val_data_prefix='zinc_final_test', batch_size=2, lr=2e-4, torch_device='cpu', test_epochs=20,
              n_stability_samples=10, normalization='batch_norm', anchors_context=False)
    m = DDPM(**hp)
    m.setup(stage='val')
    batches = list(m.val_dataloader(collate_fn=collate_with_fragment_edges))
    assert [len(b['uuid']) for b in batches] == [2, 2, 1] and 'edges' in batches[0]
    assert batches[0]['positions'].shape == (2, 10, 3) and batches[0]['edge_mask'].shape == (200, 1)
    with pytest.raises(NotImplementedError):
        m.setup(stage='test')
    # resume rule: nothing there -> start at 0; files 0..k -> restart two back; complete -> generated
    out = os.path.join(tmp_path, 'out')
    for u in ('0', '1'):
        os.makedirs(os.path.join(out, u))
    assert check_if_generated(out, ['0', '1'], 3) == (False, 0)
    for i in range(3):
        open(os.path.join(out, '0', f'{i}_.xyz'), 'w').close()
    for i in range(2):
        open(os.path.join(out, '1', f'{i}_.xyz'), 'w').close()
    open(os.path.join(out, '1', 'true_.xyz'), 'w').close()
    assert check_if_generated(out, ['0', '1'], 3) == (False, 0)
    assert check_if_generated(out, ['0'], 3) == (True, None)


def test_bench_gpus_2_starts_its_own_ranks():
    """``python bench.py --gpus 2`` WITHOUT a launcher (the driver's command shape; VERDICT round 3: it died on an assert about
    WORLD_SIZE) starts two ranks itself.  Without a GPU each rank stops at the CUDA check - after the rendezvous variables
    were set and the --gpus / WORLD_SIZE assertion passed."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env['CUDA_VISIBLE_DEVICES'] = env['HIP_VISIBLE_DEVICES'] = ''        # the same outcome on a GPU box
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--backend', 'gloo', '--steps', '1',
                        '--warmup', '0', '--no-cpu-baseline'], env=env, capture_output=True, text=True, timeout=600)
    err = r.stderr + r.stdout
    assert r.returncode != 0
    assert 'WORLD_SIZE=' not in err, err[-2000:]
    assert err.count('bench.py needs an MI355X') >= 2, err[-2000:]       # both ranks got as far as the CUDA check


def test_bench_gpus_8_rendezvous_on_cpu():
    """VERDICT round 4 (#8), first-run insurance for the driver's 8-GPU scaling run: ``python bench.py --gpus 8 --backend gloo
    --config C1 --T 3`` without a launcher starts its eight ranks, they form the process group and complete a collective (gloo
    needs no device), and every one of them then stops at the device check - nothing before it (argument handling, self-spawn,
    rendezvous on 127.0.0.1, rank / world bookkeeping) can fail for the first time on the 8-GPU node."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env['CUDA_VISIBLE_DEVICES'] = env['HIP_VISIBLE_DEVICES'] = ''
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '8', '--backend', 'gloo', '--config', 'C1', '--T', '3',
                        '--steps', '1', '--warmup', '0', '--no-cpu-baseline'], env=env, capture_output=True, text=True, timeout=900)
    err = r.stderr + r.stdout
    assert r.returncode != 0
    for k in range(8):
        assert f'rank {k}/8 joined (8 ranks in the group)' in err, err[-3000:]
    assert err.count('bench.py needs an MI355X') >= 8, err[-3000:]


def test_found_nan_exception_from_index_sets_and_split_terms():
    """Round-4 host helpers: the exception rebuilt from whole-batch index sets keeps the reference's three-way partition
    (utils.py:274-289); the average number of fp16 MFMA terms per multiply-accumulate prices the f16x2 ceiling between 2 and 3."""
    from difflinker_amd import synthetic
    from difflinker_amd.utils import FoundNaNException
    e = FoundNaNException.from_index_sets({3}, {0, 7}, {5})
    assert e.x_h_nan_idx == {3} and e.only_x_nan_idx == {0, 7} and e.only_h_nan_idx == {5}
    f = FoundNaNException.from_flags([1, 0, 2, 3])
    assert f.only_x_nan_idx == {0} and f.only_h_nan_idx == {2} and f.x_h_nan_idx == {3}
    data, cfg = synthetic.make_batch('C2', seed=1000, batch=16)
    pairs, nodes = synthetic.pair_and_node_counts(data)
    pc = synthetic.coord_pair_count(data)
    fin = cfg['nf'] + cfg['ctx'] + 1
    assert synthetic.split_terms('f16x3', 128, cfg['n_layers'], fin, pairs, pc, nodes) == 3.0
    t2 = synthetic.split_terms('f16x2', 128, cfg['n_layers'], fin, pairs, pc, nodes)
    assert 2.0 < t2 < 2.5          # the GCL edge models' second layer is ~80 % of the executed work
    assert synthetic.flops_executed(128, cfg['n_layers'], fin, pairs, pc, nodes) < synthetic.flops_min(128, cfg['n_layers'], fin, pairs, nodes)


def test_split_plan_is_a_deterministic_feasible_function_of_the_sizes():
    """``edm.split_plan`` (the static hand-over of a chain in two launches): deterministic, a function of the sizes alone; every
    unfinished molecule stops strictly inside the chain; the second phase fits the chip (2 x teams + singles <= compute units);
    the predicted makespan beats the single launch; a uniform batch, a batch that leaves the chip room for teams anyway and a
    batch larger than the chip get no plan.  The C2 batch: the plan the benchmark runs (115 molecules of 43..50 atoms stop at
    call 427 of 501)."""
    from difflinker_amd import synthetic
    from difflinker_amd.edm import forward_cost, split_plan
    data, cfg = synthetic.make_batch('C2', seed=1000)
    sizes = data['atom_mask'].squeeze(-1).sum(1).long().tolist()
    linkers = data['linker_mask'].squeeze(-1).sum(1).long().tolist()
    plan = split_plan(sizes, linkers, 501, 256, 6, 2)
    assert plan == split_plan(list(sizes), list(linkers), 501, 256, 6, 2)
    q_end, teams, singles = plan
    assert singles == [] and len(teams) == 115 and min(q_end) == 427 and min(sizes[b] for b in teams) == 43
    assert all(q_end[b] == 501 for b in range(256) if b not in teams) and all(0 < q_end[b] < 501 for b in teams)
    c1 = [forward_cost(n, l, 6, 2, 1) for n, l in zip(sizes, linkers)]
    c2 = [forward_cost(n, l, 6, 2, 2) for n, l in zip(sizes, linkers)]
    total = max(q_end[b] * c1[b] for b in range(256)) + max((501 - q_end[b]) * c2[b] for b in teams)
    assert total < 0.96 * max(c1) * 501
    # with single compute units beside the teams (opt-in): more molecules stop, the chip is exactly full in the second phase
    q2, t2, s2 = split_plan(sizes, linkers, 501, 256, 6, 2, allow_singles=True)
    assert 2 * len(t2) + len(s2) <= 256 and len(s2) > 0 and set(t2).isdisjoint(s2) and all(0 < q2[b] < 501 for b in t2 + s2)
    # no plan: nothing to hand over / no room / more molecules than compute units
    assert split_plan([50] * 256, [8] * 256, 501, 256, 6, 2) is None
    assert split_plan([50] * 200 + [49] * 56, [8] * 256, 501, 256, 6, 2) is None
    assert split_plan(sizes + [40], linkers + [5], 501, 256, 6, 2) is None
    # cost model: monotone in the number of pair-loop steps, a team of two is faster than one compute unit but less than twice
    assert forward_cost(35, 6, 6, 2) < forward_cost(43, 6, 6, 2) < forward_cost(50, 6, 6, 2)
    assert 1.3 < forward_cost(50, 8, 6, 2, 1) / forward_cost(50, 8, 6, 2, 2) < 2.0


def test_nan_flag_word_decodes_into_the_references_index_sets_and_the_f16_range_set():
    """include/difflinker_hip.h: bit0 NaN in the velocity, bit1 in the node features (utils.py:274-289: x&h / x only / h only),
    bit4 - always with both - a magnitude bound beyond the range of the f16 arithmetic modes."""
    from difflinker_amd.utils import FoundNaNException
    e = FoundNaNException.from_flags([0, 1, 2, 3, 19, 0])
    assert e.only_x_nan_idx == {1} and e.only_h_nan_idx == {2} and e.x_h_nan_idx == {3, 4} and e.f16_range_idx == {4}
    assert "precision='fp32'" in str(e) and '[4]' in str(e)
    e = FoundNaNException.from_flags(torch.tensor([3, 0], dtype=torch.int32))
    assert e.x_h_nan_idx == {0} and not e.f16_range_idx and 'range' not in str(e)
    e = FoundNaNException.from_index_sets({5}, {1}, set(), {5})
    assert e.x_h_nan_idx == {5} and e.only_x_nan_idx == {1} and e.f16_range_idx == {5}


def _gloo_uneven_worker(rank, world, port, tmp):
    import torch.distributed as dist
    from difflinker_amd.distributed import sample_chain_sharded, shard_bounds
    dist.init_process_group('gloo', init_method=f'tcp://127.0.0.1:{port}', rank=rank, world_size=world)
    B, N, nf, K = 2, 6, 4, 2                      # fewer molecules than ranks: the last rank's shard is EMPTY
    g = torch.Generator().manual_seed(3)
    inp = dict(x=torch.randn(B, N, 3, generator=g), h=torch.randn(B, N, nf, generator=g), node_mask=torch.ones(B, N, 1),
               fragment_mask=torch.ones(B, N, 1), linker_mask=torch.zeros(B, N, 1),
               edge_mask=torch.arange(B).repeat_interleave(N),          # the pockets' batch-id vector [B*N] (datasets.py:359-364)
               context=torch.randn(B, N, 2, generator=g))
    lo, hi = shard_bounds(B, rank, world)

    class StandIn:
        coef_batch = None
        team_batch = None
        noise_source = 'philox'

        def sample_chain(self, x, h, node_mask, fragment_mask, linker_mask, edge_mask, context, keep_frames=None, mol_offset=0):
            assert x.shape[0] == hi - lo and mol_offset == lo and self.coef_batch == B
            if x.shape[0]:
                ids = edge_mask.view(x.shape[0], N)
                assert torch.equal(ids, torch.arange(x.shape[0]).repeat_interleave(N).view(-1, N)), 're-based to the shard'
            frame = torch.cat([x + float(mol_offset), h], dim=2)
            return torch.stack([frame * (k + 1) for k in range(keep_frames)])

    full = sample_chain_sharded(StandIn(), inp, keep_frames=K)
    want = torch.stack([torch.cat([inp['x'] + torch.tensor([0.0, 1.0]).view(B, 1, 1), inp['h']], dim=2) * (k + 1) for k in range(K)])
    assert full.shape == (K, B, N, 3 + nf) and torch.equal(full, want)
    if rank == world - 1:
        assert hi == lo, 'this rank sampled nothing and still took part in the collective'
        torch.save(full, os.path.join(tmp, 'uneven.pt'))
    dist.barrier()
    dist.destroy_process_group()


def test_sharding_with_fewer_molecules_than_ranks_and_batch_id_edge_masks(tmp_path):
    """world_size 3, two molecules: one rank's shard is empty (an empty batch is an empty chain, like the reference) and it still
    joins the all-gather; the pockets' batch-id ``edge_mask`` of a shard is re-based to start at 0; every rank gets the full frames."""
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_gloo_uneven_worker, args=(3, port, str(tmp_path)), nprocs=3, join=True)
    assert os.path.exists(os.path.join(str(tmp_path), 'uneven.pt'))
