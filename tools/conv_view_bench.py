"""Time the patch-matrix (ia_conv_nhwc_*) and patch-matrix-free (ia_conv3x3_padded_*) 3x3 grouped convolutions on the
ECA-NFNet-L0 stage shapes (800x800 input, 2B = 32 images): python tools/conv_view_bench.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from item_alignment_amd import _lib
from item_alignment_amd._lib import check, stream_ptr

lib = _lib.load()
dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 32


def timed(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for (H, groups) in ((200, 1), (100, 2), (50, 6), (25, 6)):
    W, C = H, groups * 64
    x = torch.randn((N, H, W, C), device=dev).bfloat16()
    dy = torch.randn((N, H, W, C), device=dev).bfloat16()
    what = (torch.randn((C, 576), device=dev) * 0.05).bfloat16()
    xp = torch.nn.functional.pad(x, (0, 0, 1, 1, 1, 1)).contiguous()
    dyp = torch.nn.functional.pad(dy, (0, 0, 1, 1, 1, 1)).contiguous()
    y, yp, dx, dxp = torch.empty_like(x), torch.empty_like(xp), torch.empty_like(x), torch.empty_like(xp)
    dwhat = torch.empty((C, 576), device=dev, dtype=torch.float32)
    wsb = lib.ia_conv_nhwc_workspace_bytes(N, H, W, C, C, 3, 1, groups)
    ws = torch.empty(wsb, device=dev, dtype=torch.uint8)
    wsb2 = lib.ia_conv3x3_padded_workspace_bytes(N, H, W, C, C, groups)
    ws2 = torch.empty(max(wsb2, 16), device=dev, dtype=torch.uint8)
    s = stream_ptr()
    flops = 2.0 * N * H * W * C * 576
    old = [timed(lambda: check(lib.ia_conv_nhwc_fwd(x.data_ptr(), what.data_ptr(), None, y.data_ptr(), N, H, W, C, C, 3, 1, groups, ws.data_ptr(), wsb, s), "f")),
           timed(lambda: check(lib.ia_conv_nhwc_bwd_data(dy.data_ptr(), what.data_ptr(), dx.data_ptr(), N, H, W, C, C, 3, 1, groups, ws.data_ptr(), wsb, s), "d")),
           timed(lambda: check(lib.ia_conv_nhwc_bwd_weight(x.data_ptr(), dy.data_ptr(), dwhat.data_ptr(), None, N, H, W, C, C, 3, 1, groups, 0, ws.data_ptr(), wsb, s), "w"))]
    new = [timed(lambda: check(lib.ia_conv3x3_padded_fwd(xp.data_ptr(), what.data_ptr(), None, yp.data_ptr(), N, H, W, C, C, groups, s), "f")),
           timed(lambda: check(lib.ia_conv3x3_padded_bwd_data(dyp.data_ptr(), what.data_ptr(), dxp.data_ptr(), N, H, W, C, C, groups, s), "d")),
           timed(lambda: check(lib.ia_conv3x3_padded_bwd_weight(xp.data_ptr(), dyp.data_ptr(), dwhat.data_ptr(), None, N, H, W, C, C, groups, ws2.data_ptr(), wsb2, s), "w"))]
    print(f"H={H} C={C} groups={groups}: " + "  ".join(f"{n} {o:.3f}->{v:.3f} ms ({flops / v / 1e9:.0f} TF)" for n, o, v in zip(("fwd", "dgrad", "wgrad"), old, new)), flush=True)
