"""Multi-GPU sampling: molecules are independent (FC edges are intra-molecule, egnn.py:456-461;
gamma scalars are per-step constants), so a sampling batch shards across ranks with no data-path
collective; only the final frames are exchanged, by ONE all-gather (RCCL over xGMI when the
process group's backend is ``nccl``; ``gloo`` in the CPU tests).

One process per GPU (``torch.distributed``); rank r takes a contiguous slice of the collated
batch.  Noise comes from the in-kernel counter-based generator keyed by the GLOBAL molecule index
(or from an explicit global bank, sliced), so the sample of molecule b does not depend on the world size - bit for bit, with
default settings (round 6): a shard pins ``EDM.coef_batch`` / ``EDM.team_batch`` to the whole batch, which also keeps its chain
in ONE launch on one compute unit per molecule - the two-launch hand-over of ``EDM.split_chain`` and the surplus teams of
``EDM.overflow_teams`` follow the sizes of the molecules at hand and are for unsharded batches only (they change the order in
which a molecule's messages are summed: ~1e-8 on the final coordinates).  The unsharded call of a batch BEYOND one GPU's compute
units runs one launch per chain too, so world = 1 / 2 / 4 / 8 sample the same bits (tests/test_gpu_round6.py, C3 shape).
"""
import time

import torch
import torch.distributed as dist


def shard_bounds(n_items, rank, world_size):
    """Contiguous, balanced [lo, hi) slice of ``n_items`` for ``rank`` (first ``n_items % world``
    ranks get one extra item)."""
    base, extra = divmod(n_items, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_sampler_inputs(inputs, rank, world_size):
    """Slice the ``EDM.sample_chain`` tensors of a collated batch for one rank.

    ``edge_mask`` is either the FC mask ``[B*N*N, 1]`` (sliced by molecule) or the pockets' batch-id
    vector ``[B*N]`` (sliced and re-based so ids start at 0)."""
    bs, n = inputs['x'].shape[0], inputs['x'].shape[1]
    lo, hi = shard_bounds(bs, rank, world_size)
    out = {}
    for k, v in inputs.items():
        if k == 'edge_mask' and v is not None:
            if v.numel() == bs * n * n:
                out[k] = v.view(bs, n * n)[lo:hi].reshape(-1, 1)
            else:
                out[k] = (v.view(bs, n)[lo:hi] - lo).reshape(-1)
        elif torch.is_tensor(v) and v.dim() >= 1 and v.shape[0] == bs:
            out[k] = v[lo:hi]
        else:
            out[k] = v
    return out, (lo, hi)


def all_gather_frames(local_chain, batch_size, group=None):
    """All-gather per-rank chains ``[K, B_r, N, D]`` into ``[K, B, N, D]`` (rank order = batch order).
    Shards may differ by one molecule: they are padded to the largest shard for the collective."""
    if not dist.is_initialized():
        return local_chain
    # (a process group of ONE rank still goes through the collective: the RCCL branch below then runs - and is tested - on a
    # single-GPU box; a job without a process group, bench.py --gpus 1, returns above)
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    sizes = [shard_bounds(batch_size, r, world) for r in range(world)]
    bmax = max(hi - lo for lo, hi in sizes)
    k, b_r, n, d = local_chain.shape
    assert b_r == sizes[rank][1] - sizes[rank][0]
    send = local_chain.new_zeros((k, bmax, n, d))
    send[:, :b_r] = local_chain
    # RCCL ('nccl') gathers device tensors in place; gloo (several ranks on ONE GPU: functional checks on a single-GPU box)
    # has no device all-gather - the frames take the detour through host memory there
    via_host = send.is_cuda and dist.get_backend(group) == 'gloo'
    timed = send.is_cuda and not via_host
    if timed:                                 # the collective's duration on the stream, for bench.py (read later: no sync here)
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record()
    else:
        t0 = time.perf_counter()
    if via_host:
        send = send.cpu()
    recv = [torch.empty_like(send) for _ in range(world)]
    dist.all_gather(recv, send.contiguous(), group=group)
    out = torch.cat([recv[r][:, :sizes[r][1] - sizes[r][0]] for r in range(world)], dim=1)
    out = out.to(local_chain.device) if via_host else out
    if timed:
        ev[1].record()
        LAST_GATHER['events'], LAST_GATHER['host_s'] = ev, None
    else:
        LAST_GATHER['events'], LAST_GATHER['host_s'] = None, time.perf_counter() - t0
    return out


# duration of the most recent all_gather_frames of this process: a pair of stream events (RCCL) or host seconds (gloo)
LAST_GATHER = {'events': None, 'host_s': None}


def last_gather_ms():
    if LAST_GATHER['events'] is not None:
        s, e = LAST_GATHER['events']
        e.synchronize()
        return s.elapsed_time(e)
    return None if LAST_GATHER['host_s'] is None else 1e3 * LAST_GATHER['host_s']


def sample_chain_sharded(edm, inputs, keep_frames=None, noise_bank=None, group=None, gather=True):
    """``edm.sample_chain`` on this rank's slice of the batch, then one all-gather of the frames.

    ``inputs``: dict with the keyword tensors of ``EDM.sample_chain`` for the FULL batch (already on
    this rank's device).  Noise: with more than one rank the draws are generated inside the kernels by the
    counter-based generator (``noise_source='philox'``, keyed by seed and GLOBAL molecule index: no rank ever
    materialises the global bank, and a sample does not depend on the world size); every rank must hold the same
    ``edm.noise_seed``.  ``edm.noise_source = 'torch'`` is honoured only when the caller hands over the global
    ``noise_bank`` explicitly (parity tests): drawing the reference's ``torch.randn`` stream for the whole batch on every
    rank costs the full bank (2.5 GB at config C3) and the full ``randn`` work per GPU.
    ``InpaintingEDM`` (edm.py:549-730) is sharded the same way (round 5): its ``1 + 2T + 2`` draws per molecule come from the
    same counter-based generator, keyed by the global molecule index; its centre-of-gravity projections are per molecule."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    bs, n = inputs['x'].shape[0], inputs['x'].shape[1]
    local, (lo, hi) = shard_sampler_inputs(inputs, rank, world)
    kw = {}
    source = getattr(edm, 'noise_source', 'torch')
    if noise_bank is not None:
        kw['noise_bank'] = (noise_bank[0][:, lo:hi].contiguous(), noise_bank[1][:, lo:hi].contiguous())
    elif source == 'philox' or world > 1:
        kw['mol_offset'] = lo                 # counter-based draws: nothing to generate or slice, any world size
        edm.noise_source = 'philox'
    pinned, pinned_team = getattr(edm, 'coef_batch', None), getattr(edm, 'team_batch', None)
    if pinned is None:
        edm.coef_batch = bs                  # per-step scalars of the WHOLE batch (see EDM.coef_batch)
    if pinned_team is None:
        edm.team_batch = bs                  # and its team size (see EDM.team_batch)
    try:
        chain = edm.sample_chain(keep_frames=keep_frames, **local, **kw)
    finally:
        edm.coef_batch = pinned
        edm.team_batch = pinned_team
        edm.noise_source = source
    return all_gather_frames(chain, bs, group) if gather else chain
