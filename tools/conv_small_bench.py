"""The zero-bordered 3x3 / stride-1 convolutions of eca_nfnet_l0 with few channels (stem conv2 / conv3, the stage-0 / stage-1 blocks'
conv2 / conv2b) against their HBM floor (each operand once): python tools/conv_small_bench.py [images]"""
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


for (H, Cin, Cout, groups, what_) in ((400, 16, 32, 1, "stem conv2"), (400, 32, 64, 1, "stem conv3"), (200, 64, 64, 1, "stage 0 conv2 / conv2b"),
                                      (100, 128, 128, 2, "stage 1 conv2b"), (50, 384, 384, 6, "stage 2 conv2b"), (25, 384, 384, 6, "stage 3 conv2b")):
    W = H
    ci = Cin // groups
    Mp = N * (H + 2) * (W + 2)
    xp = torch.randn((Mp, Cin), device=dev).bfloat16()
    dyp = torch.randn((Mp, Cout), device=dev).bfloat16()
    what = (torch.randn((Cout, 9 * ci), device=dev) * 0.05).bfloat16()
    yp, dxp = torch.empty((Mp, Cout), device=dev, dtype=torch.bfloat16), torch.empty((Mp, Cin), device=dev, dtype=torch.bfloat16)
    dwhat = torch.empty((Cout, 9 * ci), device=dev, dtype=torch.float32)
    wsb = lib.ia_conv3x3_padded_workspace_bytes(N, H, W, Cin, Cout, groups)
    ws = torch.empty(max(wsb, 16), device=dev, dtype=torch.uint8)
    s = stream_ptr()
    direct = bool(lib.ia_conv3x3_direct_supported(Cin, Cout, groups))     # the data gradient = the direct kernel on the flipped bank (co <-> ci)
    what_t = torch.empty((Cin, 9 * (Cout // groups)), device=dev, dtype=torch.bfloat16)
    if direct:
        check(lib.ia_conv3x3_flip_weights(what.data_ptr(), what_t.data_ptr(), Cin, Cout, groups, s), "flip")

    def dgrad():
        if direct:      # as models/nfnet.py: flip (a few us, part of the timed call) + direct kernel
            check(lib.ia_conv3x3_flip_weights(what.data_ptr(), what_t.data_ptr(), Cin, Cout, groups, s), "flip")
            check(lib.ia_conv3x3_padded_bwd_data_t(dyp.data_ptr(), what_t.data_ptr(), dxp.data_ptr(), N, H, W, Cin, Cout, groups, s), "dt")
        else:
            check(lib.ia_conv3x3_padded_bwd_data(dyp.data_ptr(), what.data_ptr(), dxp.data_ptr(), N, H, W, Cin, Cout, groups, s), "d")
    t = [timed(lambda: check(lib.ia_conv3x3_padded_fwd(xp.data_ptr(), what.data_ptr(), None, yp.data_ptr(), N, H, W, Cin, Cout, groups, s), "f")),
         timed(dgrad),
         timed(lambda: check(lib.ia_conv3x3_padded_bwd_weight(xp.data_ptr(), dyp.data_ptr(), dwhat.data_ptr(), None, N, H, W, Cin, Cout, groups, ws.data_ptr(), wsb, s), "w"))]
    gb = Mp * (Cin + Cout) * 2 / 1e9
    flops = 2.0 * N * H * W * Cout * 9 * ci
    print(f"{what_:24s} {H}x{H} {Cin:3d}->{Cout:3d} g{groups}: " + "  ".join(f"{n} {v:.3f} ms ({gb / v:5.2f} TB/s, {flops / v / 1e9:4.0f} TF)" for n, v in zip(("fwd", "dgrad", "wgrad"), t)) +
          f"   floor {gb / 5.5 * 1e0:.3f} ms each at 5.5 TB/s", flush=True)
