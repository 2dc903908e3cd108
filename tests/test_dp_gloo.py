"""Data-parallel path on CPU: two gloo processes run the bucketed gradient reducer and the rank-sharded sampler
(the RCCL path on the GPUs is the same code with backend "nccl")."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from item_alignment_amd import dist as iadist
    r, w, _ = iadist.init_from_env("cpu")
    assert (r, w) == (rank, world)
    torch.manual_seed(0)
    n = 10_000
    flat = torch.full((n,), float(rank + 1))
    params, off = [], 0
    for m in (1000, 4000, 2500, 2500):
        p = torch.nn.Parameter(torch.zeros(m)); params.append((p, off, m)); off += m
    red = iadist.GradBucketReducer(flat, params, bucket_bytes=3000 * 4)
    for p, _, _ in reversed(params):                     # backward order: last parameter first
        red.grads_ready([p])
    scale = red.finish()
    ok = bool(torch.allclose(flat, torch.full((n,), 3.0))) and scale == 0.5
    # second round re-uses the reducer (state reset), with one parameter never reported (frozen)
    flat.fill_(float(rank + 1))
    red.grads_ready([params[3][0]])
    red.finish()
    ok = ok and bool(torch.allclose(flat, torch.full((n,), 3.0)))
    # gradient accumulation: micro-step 1 (disarmed) only fills the arena, micro-step 2 (armed) adds to it and reduces once
    flat.fill_(float(rank + 1))
    red.armed = False
    for p, _, _ in reversed(params):
        red.grads_ready([p])
    ok = ok and not any(red.launched)
    flat.add_(float(rank + 1))
    red.armed = True
    for p, _, _ in reversed(params):
        red.grads_ready([p])
    red.finish()
    ok = ok and bool(torch.allclose(flat, torch.full((n,), 6.0)))
    shard = iadist.shard_indices(101, rank, world, epoch_seed=5)
    gathered = [torch.zeros_like(shard) for _ in range(world)]
    dist.all_gather(gathered, shard)
    allidx = torch.cat(gathered)
    ok = ok and len(set(allidx.tolist())) == len(allidx) == 100
    out[rank] = ok
    dist.destroy_process_group()


def test_bucketed_allreduce_two_ranks():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    assert dict(out) == {0: True, 1: True}
