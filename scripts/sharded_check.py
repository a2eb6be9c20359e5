"""Functional test of the multi-GPU path on whatever devices the box has (W processes on ONE GPU when that is
all there is):  python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
    scripts/sharded_check.py [--backend nccl|gloo]
Every rank samples its shard of one logical C2-like batch through ``distributed.sample_chain_sharded`` (in-kernel Philox
noise, one all-gather of the final frame); rank 0 compares the gathered chain with the unsharded run: must be BITWISE equal."""
import argparse
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument('--backend', default='nccl')
ap.add_argument('--batch', type=int, default=48)
ap.add_argument('--T', type=int, default=40)
a = ap.parse_args()
rank, world, local = int(os.environ['RANK']), int(os.environ['WORLD_SIZE']), int(os.environ['LOCAL_RANK'])
ndev = torch.cuda.device_count()
device = torch.device('cuda', local % ndev)              # ranks share a device when there are fewer GPUs than ranks
torch.cuda.set_device(device)
os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
kw = {'device_id': device} if a.backend == 'nccl' and ndev >= world else {}
dist.init_process_group(a.backend, rank=rank, world_size=world, **kw)

from difflinker_amd import Dynamics, EDM, synthetic
from difflinker_amd.distributed import sample_chain_sharded

data, cfg = synthetic.make_batch('C2', seed=7, batch=a.batch)
inp = {k: v.to(device) for k, v in synthetic.sampler_inputs(data).items()}
torch.manual_seed(0)
dyn = Dynamics(n_dims=3, in_node_nf=cfg['nf'], context_node_nf=cfg['ctx'], hidden_nf=128, n_layers=2, norm_constant=1e-6)
edm = EDM(dyn, in_node_nf=cfg['nf'], n_dims=3, timesteps=500, noise_schedule='polynomial_2', noise_precision=1e-5,
          loss_type='l2', norm_values=[1, 4, 10]).to(device)
edm.T = a.T
edm.noise_source = 'philox'
edm.noise_seed = 11
edm.split_chain = edm.overflow_teams = False       # one launch per chain: the setting under which a sample is BITWISE independent of the split
got = sample_chain_sharded(edm, inp, keep_frames=2)
torch.cuda.synchronize()
if world == 1:
    # a single rank never reaches the collective inside all_gather_frames: push the frames through one RCCL / gloo
    # all-gather anyway, so that the backend's init + collective path has run on this software stack
    recv = [torch.empty_like(got)]
    dist.all_gather(recv, got.contiguous())
    torch.cuda.synchronize()
    assert torch.equal(recv[0], got)
    print(f'sharded_check: backend={a.backend} world_size=1: all_gather of the frames through the backend ok', flush=True)
dist.barrier()
if rank == 0:
    edm.noise_seed = 11
    edm.coef_batch = None
    want = edm.sample_chain(keep_frames=2, **inp)
    same = torch.equal(got, want)
    print(f'sharded_check: backend={a.backend} world_size={world} devices={ndev} batch={a.batch} T={a.T} '
          f'gathered chain {tuple(got.shape)} == unsharded: {"BITWISE EQUAL" if same else "DIFFERENT"} '
          f'(max |diff| {float((got - want).abs().max()):.3e})', flush=True)
    assert same
dist.barrier()
dist.destroy_process_group()
