"""BatchNorm+ReLU forward / backward (ia_bn_act_*) bandwidth on the resnetv2_50 @800x800 activation shapes: python tools/bn_bench.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from item_alignment_amd import _lib
from item_alignment_amd._lib import check, stream_ptr
lib = _lib.load()
dev = torch.device("cuda:0")
def timed(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for rows, C in [(640000, 256), (640000, 64), (160000, 512), (40000, 1024), (10000, 2048)]:
    seg = 2
    x = torch.randn((rows, C), device=dev).bfloat16(); dy = torch.randn((rows, C), device=dev).bfloat16(); ex = torch.randn((rows, C), device=dev).bfloat16()
    y = torch.empty_like(x); dx = torch.empty_like(x)
    ga, be = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    mean, rstd = torch.empty((seg, C), device=dev), torch.empty((seg, C), device=dev)
    dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    wsb = lib.ia_bn_act_workspace_bytes(rows, C, seg); ws = torch.empty(wsb, device=dev, dtype=torch.uint8)
    f = timed(lambda: check(lib.ia_bn_act_fwd(x.data_ptr(), ga.data_ptr(), be.data_ptr(), rm.data_ptr(), rv.data_ptr(), y.data_ptr(), mean.data_ptr(), rstd.data_ptr(), rows, C, seg, 1e-5, 0.1, 1, 1, ws.data_ptr(), wsb, stream_ptr()), "f"))
    b = timed(lambda: check(lib.ia_bn_act_bwd(dy.data_ptr(), x.data_ptr(), ga.data_ptr(), be.data_ptr(), mean.data_ptr(), rstd.data_ptr(), ex.data_ptr(), dx.data_ptr(), dg.data_ptr(), db.data_ptr(), rows, C, seg, 1, 1, ws.data_ptr(), wsb, stream_ptr()), "b"))
    mb = rows * C * 2 / 1e6
    print(f"rows {rows} C {C}: fwd {f*1e3:.0f} us ({3*mb/f/1e6:.1f} TB/s)  bwd {b*1e3:.0f} us ({6*mb/b/1e6:.1f} TB/s)", flush=True)
