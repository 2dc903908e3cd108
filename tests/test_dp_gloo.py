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
        red.grads_ready([p], final=True)
    ok0 = all(red.launched)                              # every bucket went out during "backward"
    scale = red.finish()
    ok = ok0 and bool(torch.allclose(flat, torch.full((n,), 3.0))) and scale == 0.5
    # a parameter whose gradient is produced by two autograd nodes (module applied twice, chunked tower): its reports are not
    # final, so its bucket must NOT be reduced before the second contribution has been added
    flat.fill_(float(rank + 1))
    red.grads_ready([params[3][0]], final=False)         # first node
    launched_early = red.launched[0]
    flat[7500:].add_(10.0 * (rank + 1))                  # second node adds to the same gradient
    red.grads_ready([params[3][0]], final=False)
    for p, _, _ in reversed(params[:3]):
        red.grads_ready([p], final=True)
    red.finish()
    want = torch.full((n,), 3.0); want[7500:] += 30.0
    ok = ok and not launched_early and bool(torch.allclose(flat, want))
    # second round re-uses the reducer (state reset), with one parameter never reported (frozen)
    flat.fill_(float(rank + 1))
    red.grads_ready([params[3][0]], final=True)
    red.finish()
    ok = ok and bool(torch.allclose(flat, torch.full((n,), 3.0)))
    # gradient accumulation: micro-step 1 (disarmed) only fills the arena, micro-step 2 (armed) adds to it and reduces once
    flat.fill_(float(rank + 1))
    red.armed = False
    for p, _, _ in reversed(params):
        red.grads_ready([p], final=True)
    ok = ok and not any(red.launched)
    flat.add_(float(rank + 1))
    red.armed = True
    for p, _, _ in reversed(params):
        red.grads_ready([p], final=True)
    red.finish()
    ok = ok and bool(torch.allclose(flat, torch.full((n,), 6.0)))
    shard = iadist.shard_indices(101, rank, world, epoch_seed=5)
    gathered = [torch.zeros_like(shard) for _ in range(world)]
    dist.all_gather(gathered, shard)
    allidx = torch.cat(gathered)
    ok = ok and len(set(allidx.tolist())) == len(allidx) == 100
    # sharded evaluation: every rank hands in the (score, label) rows it produced -- the counts may differ (dropped samples)
    # and a shard may be empty -- and gets the concatenation of all ranks back
    import numpy as np
    whole = np.stack([np.arange(11, dtype=np.float64) * 0.5, np.arange(11, dtype=np.float64) % 2], 1)
    mine = whole[rank::world]
    if rank == 1:
        mine = mine[:-2]                      # two samples of rank 1 failed to load
    back = iadist.gather_rows(mine)
    want = np.concatenate([whole[0::world], whole[1::world][:-2]], 0)
    ok = ok and back.shape == want.shape and bool((back == want).all())
    back = iadist.gather_rows(whole[:3] if rank == 0 else None)      # rank 1 came back with nothing at all
    ok = ok and back.shape == (3, 2) and bool((back == whole[:3]).all())
    back = iadist.gather_rows(None)
    ok = ok and back.shape[0] == 0
    # plain torch models (TextCNN): gradient averaging without an arena
    lin = torch.nn.Linear(3, 2)
    for p in lin.parameters():
        p.grad = torch.full_like(p, float(rank + 1))
    iadist.all_reduce_grads(lin, world)
    ok = ok and all(bool(torch.allclose(p.grad, torch.full_like(p, 1.5))) for p in lin.parameters())
    out[rank] = ok
    dist.destroy_process_group()


def test_bucketed_allreduce_two_ranks():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    assert dict(out) == {0: True, 1: True}
