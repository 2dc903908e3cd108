"""Development harness for the LDS-resident attention kernels: parity against fp32 torch on the same bf16 inputs and timing next
to the streaming kernels.  usage: python tools/attn_dev.py [check] [time]"""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from item_alignment_amd import ops, _lib
from item_alignment_amd.ops import check, ptr, stream_ptr

dev = torch.device("cuda:0")
QS = 0.125 * math.log2(math.e)
LN2 = math.log(2.0)


def ref(qkv_s, B, L, nh, mask, dctx=None):
    """softmax over ln2 * (q' . k): q' is the pre-scaled query the resident kernels take"""
    H = nh * 64
    t = qkv_s.float().view(B, L, 3, nh, 64).requires_grad_(dctx is not None)
    q, k, v = t[:, :, 0].transpose(1, 2), t[:, :, 1].transpose(1, 2), t[:, :, 2].transpose(1, 2)
    s = q @ k.transpose(-1, -2) * LN2
    if mask is not None:
        s = s + (1.0 - mask.float())[:, None, None, :] * torch.finfo(torch.float32).min
    p = torch.softmax(s, -1)
    ctx = (p @ v).transpose(1, 2).reshape(B * L, H)
    if dctx is None:
        return ctx, None
    ctx.backward(dctx.float())
    g = t.grad.reshape(B * L, 3, H).clone()
    g[:, 0] *= QS      # gradient with respect to the unscaled q
    return ctx.detach(), g.reshape(B * L, 3 * H)


def prescale(qkv, H):
    s = qkv.clone()
    s.view(-1, 3, H)[:, 0] = (qkv.view(-1, 3, H)[:, 0].float() * QS).to(torch.bfloat16)
    return s


def fwd_res(qkv_s, B, L, nh, mask=None, drop_p=0.0, seed=0):
    lib = _lib.load()
    H = nh * 64
    out = torch.empty((B * L, H), device=dev, dtype=torch.bfloat16)
    nlse = torch.empty((B, nh, L), device=dev, dtype=torch.float32)
    base = qkv_s.data_ptr()
    check(lib.ia_attn_fwd_res(base, base + 2 * H, base + 4 * H, 3 * H, ptr(mask), out.data_ptr(), H, nlse.data_ptr(), B, nh, L, drop_p, seed,
                              stream_ptr()), "ia_attn_fwd_res")
    return out, nlse


def bwd_res(qkv_s, ctx, dctx, nlse, B, L, nh, mask=None, drop_p=0.0, seed=0):
    lib = _lib.load()
    H = nh * 64
    dqkv = torch.empty_like(qkv_s)
    delta = torch.empty((B, nh, L), device=dev, dtype=torch.float32)
    base, dbase = qkv_s.data_ptr(), dqkv.data_ptr()
    check(lib.ia_attn_bwd_res(base, base + 2 * H, base + 4 * H, 3 * H, ptr(mask), ctx.data_ptr(), dctx.data_ptr(), H, nlse.data_ptr(),
                              delta.data_ptr(), dbase, dbase + 2 * H, dbase + 4 * H, 3 * H, B, nh, L, 0.125, drop_p, seed, stream_ptr()),
          "ia_attn_bwd_res")
    return dqkv


def rel(a, b):
    return ((a.float() - b.float()).abs().max() / b.float().abs().max().clamp_min(1e-6)).item()


def do_check(bwd):
    torch.manual_seed(0)
    ok = True
    for B, L, nh, masked in [(2, 20, 1, True), (3, 64, 2, False), (2, 255, 4, True), (2, 510, 16, True), (2, 577, 12, False), (1, 129, 2, True),
                             (2, 33, 1, False), (1, 608, 2, True), (2, 197, 3, False), (3, 96, 2, True)]:
        H = nh * 64
        qkv = (torch.randn((B * L, 3 * H), device=dev)).to(torch.bfloat16)
        dctx = torch.randn((B * L, H), device=dev).to(torch.bfloat16)
        mask = None
        if masked:
            lens = torch.tensor([max(1, L - 37 * (i + 1)) for i in range(B)])
            mask = (torch.arange(L)[None, :] < lens[:, None]).to(torch.uint8).to(dev)
            if L > 40:
                mask[0, 17] = 0
        qs = prescale(qkv, H)
        ctx, nlse = fwd_res(qs, B, L, nh, mask)
        r, g = ref(qs, B, L, nh, mask, dctx if bwd else None)
        e = rel(ctx, r)
        line = f"B={B} L={L} nh={nh} masked={masked}: fwd rel err {e:.2e}"
        ok &= e < 2e-2
        if bwd:
            dqkv = bwd_res(qs, ctx, dctx, nlse, B, L, nh, mask)
            for i, name in enumerate("qkv"):
                e = rel(dqkv.view(B * L, 3, H)[:, i], g.view(B * L, 3, H)[:, i])
                line += f"  d{name} {e:.2e}"
                ok &= e < 3e-2
        print(line, flush=True)
    # a spiked key forces the re-basing branch at a late tile
    B, L, nh = 1, 577, 2
    H = nh * 64
    qkv = torch.randn((B * L, 3 * H), device=dev).to(torch.bfloat16)
    qkv.view(L, 3, H)[500, 1] = qkv.view(L, 3, H)[7, 0] * 6
    qs = prescale(qkv, H)
    ctx, nlse = fwd_res(qs, B, L, nh)
    r, _ = ref(qs, B, L, nh, None)
    e = rel(ctx, r)
    print(f"spiked key: fwd rel err {e:.2e}")
    ok &= e < 2e-2
    # dropout: same keep stream as the streaming kernels
    B, L, nh = 2, 255, 4
    H = nh * 64
    qkv = torch.randn((B * L, 3 * H), device=dev).to(torch.bfloat16)
    qs = prescale(qkv, H)
    a, _ = fwd_res(qs, B, L, nh, drop_p=0.1, seed=123)
    b_, _ = ops.attn_fwd(qkv, B, L, nh, drop_p=0.1, seed=123)
    e = rel(a, b_)
    print(f"dropout vs streaming kernel (same seed): rel diff {e:.2e}")
    ok &= e < 3e-2
    print("CHECK", "OK" if ok else "FAILED")
    return ok


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def do_time(bwd):
    for B, L, nh, masked in [(256, 255, 16, False), (256, 255, 16, True), (64, 510, 16, False), (256, 577, 12, False)]:
        H = nh * 64
        qkv = torch.randn((B * L, 3 * H), device=dev).to(torch.bfloat16)
        qs = prescale(qkv, H)
        mask = None
        if masked:   # bench-like lengths: 141 of 255 on average
            g = torch.Generator().manual_seed(1)
            lens = torch.randint(30, 253, (B,), generator=g)
            mask = (torch.arange(L)[None, :] < lens[:, None]).to(torch.uint8).to(dev)
        fl = 4 * L * L * 64 * B * nh
        for drop in (0.0, 0.1):
            t0 = timeit(lambda: ops.attn_fwd(qkv, B, L, nh, key_mask=mask, drop_p=drop, seed=1))
            t1 = timeit(lambda: fwd_res(qs, B, L, nh, mask, drop_p=drop, seed=1))
            print(f"fwd B={B} L={L} nh={nh} masked={masked} drop={drop}: streaming {t0*1e6:7.1f} us ({fl/t0/1e12:6.1f} TF)   resident {t1*1e6:7.1f} us "
                  f"({fl/t1/1e12:6.1f} TF)", flush=True)
            if bwd:
                ctx, lse = ops.attn_fwd(qkv, B, L, nh, key_mask=mask, drop_p=drop, seed=1)
                d = torch.randn_like(ctx)
                t0 = timeit(lambda: ops.attn_bwd(qkv, ctx, d, lse, B, L, nh, key_mask=mask, drop_p=drop, seed=1))
                ctx2, nlse = fwd_res(qs, B, L, nh, mask, drop_p=drop, seed=1)
                t1 = timeit(lambda: bwd_res(qs, ctx2, d, nlse, B, L, nh, mask, drop_p=drop, seed=1))
                print(f"bwd B={B} L={L} nh={nh} masked={masked} drop={drop}: streaming {t0*1e6:7.1f} us ({2.5*fl/t0/1e12:6.1f} TF)   resident "
                      f"{t1*1e6:7.1f} us ({2.5*fl/t1/1e12:6.1f} TF)", flush=True)


if __name__ == "__main__":
    which = sys.argv[1:] or ["check", "time"]
    bwd = "nobwd" not in which
    if "check" in which:
        do_check(bwd)
    if "time" in which:
        do_time(bwd)
