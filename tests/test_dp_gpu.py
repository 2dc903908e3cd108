"""Model-level data-parallel equivalence on the GPU: two ranks, each with half of the pair batch, must produce the loss and the
parameter update of one process with the whole batch, for several optimiser steps (the reference loss is a batch mean, so equal
shards + summed gradients scaled by 1/world reproduce the global-batch gradient).

With two or more GPUs visible the ranks use one device each over RCCL (backend "nccl").  On a one-GPU box both ranks share
cuda:0 over gloo (RCCL refuses duplicate devices): the reducer, the sharding, the bucket launches from backward and the
grad-scale fold are exercised identically, only the transport differs.  PKGM one_tower is included because its KG embedding
reports the same parameters from two autograd nodes (the case that must not start a bucket early)."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

from golden_util import load_case, weights

pytestmark = pytest.mark.gpu
STEPS = 3


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _build(kind):
    import item_alignment_amd.models as M
    from test_models_gpu import cfg_of
    case = load_case({"roberta_two_tower": "roberta_two_tower_ce", "pkgm_one_tower": "pkgm_one_tower"}[kind])
    cfg = cfg_of(case)
    cfg.hidden_dropout_prob = 0.0
    cfg.attention_probs_dropout_prob = 0.0           # train mode without dropout: deterministic, rank-independent
    model = (M.RobertaTwoTower if kind == "roberta_two_tower" else M.PKGMOneTower)(cfg)
    model.load_state_dict(weights(case), strict=False)
    # the fixture's pairs twice (2 x 8 two-tower, 2 x 3 PKGM), with the second copy's labels flipped so the shards differ in content
    i = case.inputs
    rep = {k: torch.cat([v, v]) for k, v in i.items()}
    rep["labels"] = torch.cat([i["labels"], 1 - i["labels"]])
    return model, rep, case


def _call(model, kind, b):
    if kind == "roberta_two_tower":
        return model(input_ids_1=b["input_ids_1"], attention_mask_1=b["attention_mask_1"], token_type_ids_1=b["token_type_ids_1"],
                     input_ids_2=b["input_ids_2"], attention_mask_2=b["attention_mask_2"], token_type_ids_2=b["token_type_ids_2"], labels=b["labels"])
    return model(input_ids=b["input_ids"], attention_mask=b["attention_mask"], token_type_ids=b["token_type_ids"], position_ids=b["position_ids"],
                 labels=b["labels"])


def _train(kind, rank, world, bf16=False):
    from item_alignment_amd import dist as iadist
    from item_alignment_amd.models import functional as Fn
    dev = torch.device("cuda", rank % torch.cuda.device_count())
    torch.cuda.set_device(dev)
    model, batch, _ = _build(kind)
    model = model.to(dev).train()
    arena = model.param_arena
    reducer = None
    if world > 1:
        iadist.broadcast_arena(arena)
        reducer = iadist.GradBucketReducer.for_arena(arena, bucket_bytes=64 << 10, bf16=bf16)     # many small buckets
        Fn.clear_grad_ready_hooks()
        Fn.register_grad_ready_hook(reducer.grads_ready)
    shard = {k: v[rank::world].to(dev) for k, v in batch.items()}
    losses, grad0 = [], None
    for step in range(STEPS):
        Fn.set_step_seed(step)
        arena.zero_grad()
        out = _call(model, kind, shard)
        out.loss.backward()
        scale = reducer.finish() if reducer is not None else 1.0
        if step == 0:
            grad0 = (arena.grad.detach().float() * scale).cpu().clone()     # what the optimiser is about to consume
        arena.adamw_step(1e-4, grad_scale=scale)
        losses.append(out.loss.detach().float().cpu())
    torch.cuda.synchronize()
    spans = [(n, o, p.numel()) for n, p, o in zip(arena.names, arena.params, arena.offsets)]
    return torch.stack(losses), arena.master.detach().float().cpu().clone(), grad0, spans


def _worker(rank, world, port, kind, bf16, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    if torch.cuda.device_count() < world:
        os.environ["IA_DP_BACKEND"] = "gloo"         # both ranks on the one GPU
    from item_alignment_amd import dist as iadist
    import importlib
    importlib.reload(iadist)                           # picks up IA_DP_BACKEND
    iadist.init_from_env("cuda")
    losses, master, grad0, _ = _train(kind, rank, world, bf16)
    out[rank] = (losses.numpy(), master.numpy(), grad0.numpy())
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("kind,bf16", [("roberta_two_tower", False), ("pkgm_one_tower", False), ("roberta_two_tower", True)])
def test_two_ranks_equal_one_rank_on_the_global_batch(gpu, kind, bf16):
    import numpy as np
    world = 2
    ref_losses, ref_master, ref_grad, spans = _train(kind, 0, 1)
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), kind, bf16, out), nprocs=world, join=True)
    l0, m0, g0 = out[0]
    l1, m1, g1 = out[1]
    assert np.array_equal(m0, m1) and np.array_equal(g0, g1), "replicas diverged"
    # the averaged gradient every rank hands to the optimiser = the gradient of the global batch, parameter by parameter
    # (bf16 activations: the shard batches accumulate in a different order than the whole batch; bf16 buckets add their rounding)
    ref_grad = ref_grad.numpy()
    gmax = np.abs(ref_grad).max()
    for name, off, n in spans:
        want, got = ref_grad[off:off + n], g0[off:off + n]
        if np.abs(want).max() < 1e-3 * gmax:
            continue                                  # numerically empty gradient (untouched rows, cancelled sums)
        err = np.abs(got - want).max() / np.abs(want).max()
        assert err < (4e-2 if bf16 else 2e-2), (name, err)
    # the global-batch loss is the mean of the two shard losses (equal shard sizes), step after step: the updates agree too
    # (a direct parameter comparison would be ill-conditioned: Adam turns a sign flip of a near-zero gradient into a full step)
    got = (l0 + l1) / 2
    assert np.allclose(got, ref_losses.numpy(), rtol=5e-3, atol=5e-4), (got, ref_losses)


def test_comm_cabi_world_size_one(gpu):
    """ia_comm_* (RCCL behind the C ABI): id, init, in-place bucket all-reduce on a stream, finalize.  A one-GPU
    box allows world size 1 only (RCCL refuses two ranks on one device): the sum over one rank is the identity, for fp32 and bf16
    buckets, and the call is stream-ordered behind the kernel that produced the bucket."""
    import ctypes as C
    from item_alignment_amd import _lib
    lib = _lib.load()
    ident = (C.c_char * 128)()
    rc = lib.ia_comm_unique_id(ident)
    assert rc == 0, lib.ia_comm_last_error()
    comm = C.c_void_p()
    assert lib.ia_comm_init(ident, 0, 1, C.byref(comm)) == 0, lib.ia_comm_last_error()
    assert comm.value
    st = torch.cuda.current_stream().cuda_stream
    g32 = torch.randn(1 << 20, device=gpu)
    want32 = g32.clone()
    assert lib.ia_comm_allreduce_bucket(comm, g32.data_ptr(), g32.numel(), 0, st) == 0, lib.ia_comm_last_error()
    g16 = torch.randn(4096 + 8, device=gpu).to(torch.bfloat16)
    want16 = g16.clone()
    assert lib.ia_comm_allreduce_bucket(comm, g16.data_ptr(), g16.numel(), 1, st) == 0, lib.ia_comm_last_error()
    torch.cuda.synchronize()
    assert torch.equal(g32, want32) and torch.equal(g16, want16)
    # argument errors are reported, not passed to RCCL
    assert lib.ia_comm_allreduce_bucket(comm, g32.data_ptr(), 0, 0, st) == -1
    assert lib.ia_comm_allreduce_bucket(comm, g32.data_ptr(), 16, 7, st) == -1
    assert lib.ia_comm_init(ident, 1, 1, C.byref(C.c_void_p())) == -1
    assert lib.ia_comm_finalize(comm) == 0
