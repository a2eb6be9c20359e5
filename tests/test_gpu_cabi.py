"""The C ABI on its own (include/difflinker_hip.h through raw ctypes; PyTorch only provides the device buffers): a host
that is not the Python drop-in gets the same numbers and the documented status codes."""
import ctypes

import pytest
import torch

from helpers import seeded_state_dict, rel_l2
from oracle import egnn_oracle
from oracle.egnn_oracle import EGNNConfig
from test_gpu_parity import ragged_inputs, FWD_TOLS

FWD_TOL = FWD_TOLS['f16x3']          # DLConfig(..., precision = 1) below: the default arithmetic

pytestmark = pytest.mark.gpu


def test_forward_through_raw_ctypes_matches_oracle_and_reports_errors():
    from difflinker_amd import _lib
    from difflinker_amd.egnn import egnn_tensor_order
    lib = _lib.load()
    nf, ctx, L = 9, 1, 2
    sd = seeded_state_dict(nf + ctx + 1, 128, L, seed=123)
    host = [sd['dynamics.' + k].contiguous() for k in egnn_tensor_order(L)]
    cfg = _lib.DLConfig(3, nf, ctx, 128, L, 2, 1, 1e-6, 100.0, 1)
    assert lib.dl_model_num_tensors(ctypes.byref(cfg)) == len(host)
    ptrs = (ctypes.c_void_p * len(host))(*[t.data_ptr() for t in host])
    model = ctypes.c_void_p()
    assert lib.dl_model_create(ctypes.byref(cfg), ptrs, len(host), ctypes.byref(model)) == 0
    try:
        inp, z, t = ragged_inputs([20, 55, 9], [4, 7, 2], nf, seed=5)
        B, N = z.shape[:2]
        d = torch.device('cuda:0')
        xh, tt = z.to(d).contiguous(), t.to(d).contiguous()
        nm = inp['node_mask'].reshape(B, N).to(torch.int8).to(d).contiguous()
        lm = inp['linker_mask'].reshape(B, N).float().to(d).contiguous()
        em = inp['edge_mask'].reshape(B, N, N).to(torch.int8).to(d).contiguous()
        cx = inp['context'].reshape(B, N, ctx).float().to(d).contiguous()
        out = torch.full((B, N, 3 + nf), float('nan'), device=d)
        flags = torch.full((B,), -1, dtype=torch.int32, device=d)
        p = lambda x: ctypes.c_void_p(x.data_ptr())          # noqa: E731
        need = lib.dl_workspace_bytes(B, 1)                  # ABI v6: the caller owns the scratch, the callee allocates nothing
        ws = torch.empty(need, dtype=torch.uint8, device=d)
        st = lib.dl_egnn_forward_fc(model, B, N, p(xh), p(tt), 0, p(nm), p(lm), p(em), p(cx), p(out), p(flags), p(ws), need, None)
        assert st == 0
        torch.cuda.synchronize()
        ref = egnn_oracle.dynamics_forward({k: v for k, v in sd.items()}, EGNNConfig(in_node_nf=nf, context_node_nf=ctx, n_layers=L),
                                           t, z, inp['node_mask'], inp['linker_mask'], inp['edge_mask'], inp['context'])
        # the suite's forward bar for the default arithmetic (tests/test_gpu_parity.FWD_TOLS['f16x3']; VERDICT round 5: 2e-5 here)
        assert rel_l2(out.cpu()[..., 3:], ref[..., 3:]) <= FWD_TOL and flags.cpu().tolist() == [0, 0, 0]
        # the coordinates the sampler consumes, x + vel, at the same bar; the raw velocity - a difference of fp32 coordinates,
        # a few ulp(|x|) of absolute error on both sides - at the north-star bar (as tests/test_gpu_parity.report does)
        assert rel_l2(z[..., :3] + out.cpu()[..., :3], z[..., :3] + ref[..., :3]) <= FWD_TOL
        assert rel_l2(out.cpu()[..., :3], ref[..., :3]) <= 1e-4
        assert float(out.cpu()[..., :3].abs().max()) > 0
        # status codes: null pointer, negative batch, a molecule with more atoms than dl_max_atoms() (flag bit 2)
        assert lib.dl_egnn_forward_fc(model, B, N, None, p(tt), 0, p(nm), p(lm), p(em), p(cx), p(out), p(flags), p(ws), need, None) == -1
        assert lib.dl_egnn_forward_fc(model, -1, N, p(xh), p(tt), 0, p(nm), p(lm), p(em), p(cx), p(out), p(flags), p(ws), need, None) == -1
        assert lib.dl_egnn_forward_fc(model, 0, N, p(xh), p(tt), 0, p(nm), p(lm), p(em), p(cx), p(out), p(flags), p(ws), need, None) == 0
        big, zb, tb = ragged_inputs([56], [3], nf, seed=6)
        xb = zb.to(d).contiguous()
        nmb = big['node_mask'].reshape(1, 56).to(torch.int8).to(d).contiguous()
        lmb = big['linker_mask'].reshape(1, 56).float().to(d).contiguous()
        emb = big['edge_mask'].reshape(1, 56, 56).to(torch.int8).to(d).contiguous()
        cxb = big['context'].reshape(1, 56, ctx).float().to(d).contiguous()
        outb = torch.empty((1, 56, 3 + nf), device=d)
        fb = torch.zeros((1,), dtype=torch.int32, device=d)
        assert lib.dl_egnn_forward_fc(model, 1, 56, p(xb), p(tb.to(d)), 0, p(nmb), p(lmb), p(emb), p(cxb), p(outb), p(fb), p(ws), need, None) == 0
        torch.cuda.synchronize()
        assert int(fb.cpu()[0]) & 4 and float(outb.abs().max()) == 0.0
    finally:
        lib.dl_model_destroy(model)
    bad = _lib.DLConfig(3, nf, ctx, 64, L, 2, 1, 1e-6, 100.0, 1)              # hidden_nf != 128
    assert lib.dl_model_create(ctypes.byref(bad), ptrs, len(host), ctypes.byref(model)) == -2
    assert lib.dl_error_string(-3).decode().startswith('molecule exceeds')


def test_team_entry_points_through_raw_ctypes():
    """ABI v6: dl_egnn_forward_fc_team / dl_team_max / dl_workspace_bytes without the Python drop-in: same numbers as
    the one-workgroup entry point to fp32 rounding, documented refusals (team size, workspace size and alignment)."""
    from difflinker_amd import _lib
    from difflinker_amd.egnn import egnn_tensor_order
    lib = _lib.load()
    nf, ctx, L = 9, 1, 2
    sd = seeded_state_dict(nf + ctx + 1, 128, L, seed=124)
    host = [sd['dynamics.' + k].contiguous() for k in egnn_tensor_order(L)]
    cfg = _lib.DLConfig(3, nf, ctx, 128, L, 2, 1, 1e-6, 100.0, 1)
    ptrs = (ctypes.c_void_p * len(host))(*[t.data_ptr() for t in host])
    model = ctypes.c_void_p()
    assert lib.dl_model_create(ctypes.byref(cfg), ptrs, len(host), ctypes.byref(model)) == 0
    try:
        inp, z, t = ragged_inputs([20, 55, 9, 31, 2], [4, 7, 2, 5, 1], nf, seed=7)
        B, N = z.shape[:2]
        d = torch.device('cuda:0')
        xh, tt = z.to(d).contiguous(), t.to(d).contiguous()
        nm = inp['node_mask'].reshape(B, N).to(torch.int8).to(d).contiguous()
        lm = inp['linker_mask'].reshape(B, N).float().to(d).contiguous()
        em = inp['edge_mask'].reshape(B, N, N).to(torch.int8).to(d).contiguous()
        cx = inp['context'].reshape(B, N, ctx).float().to(d).contiguous()
        p = lambda x: ctypes.c_void_p(x.data_ptr())          # noqa: E731
        assert lib.dl_team_max(B) == 8
        need = max(lib.dl_workspace_bytes(B, team) for team in (1, 2, 4, 8))
        ws = torch.empty(need + 16, dtype=torch.uint8, device=d)
        outs = {}
        for team in (1, 2, 4, 8):
            out = torch.full((B, N, 3 + nf), float('nan'), device=d)
            flags = torch.full((B,), -1, dtype=torch.int32, device=d)
            st = lib.dl_egnn_forward_fc_team(model, B, N, p(xh), p(tt), 0, p(nm), p(lm), p(em), p(cx), p(out), p(flags),
                                             team, p(ws), lib.dl_workspace_bytes(B, team), None)
            torch.cuda.synchronize()
            assert st == 0 and flags.cpu().tolist() == [0] * B
            outs[team] = out.cpu()
        ref = egnn_oracle.dynamics_forward({k: v for k, v in sd.items()}, EGNNConfig(in_node_nf=nf, context_node_nf=ctx, n_layers=L),
                                           t, z, inp['node_mask'], inp['linker_mask'], inp['edge_mask'], inp['context'])
        for team in (1, 2, 4, 8):
            assert rel_l2(outs[team][..., 3:], ref[..., 3:]) <= FWD_TOL
        out = torch.empty((B, N, 3 + nf), device=d)
        flags = torch.zeros((B,), dtype=torch.int32, device=d)
        args = (model, B, N, p(xh), p(tt), 0, p(nm), p(lm), p(em), p(cx), p(out), p(flags))
        need4, need1 = lib.dl_workspace_bytes(B, 4), lib.dl_workspace_bytes(B, 1)
        assert lib.dl_egnn_forward_fc_team(*args, 3, p(ws), need, None) == -1                 # team must be 1, 2, 4 or 8
        assert lib.dl_egnn_forward_fc_team(*args, 4, p(ws), need4 - 1, None) == -1            # workspace too small
        assert lib.dl_egnn_forward_fc_team(*args, 4, None, need4, None) == -1                 # no workspace
        assert lib.dl_egnn_forward_fc_team(*args, 4, ctypes.c_void_p(ws.data_ptr() + 4), need4, None) == -1   # not 16-byte aligned
        assert lib.dl_egnn_forward_fc_team(*args, 1, None, 0, None) == -1                     # ABI v6: team 1 needs its scratch too
        assert lib.dl_egnn_forward_fc(*args, p(ws), need1 - 1, None) == -1
        assert lib.dl_egnn_forward_fc(*args, p(ws), need1, None) == 0                         # the callee allocates nothing
        torch.cuda.synchronize()
        assert flags.cpu().tolist() == [0] * B and rel_l2(out.cpu()[..., 3:], ref[..., 3:]) <= FWD_TOL
    finally:
        lib.dl_model_destroy(model)
