"""ia_ln_bwd2_rows at the bench's text shape (130 560 x 1024, dropout 0.1) with and without the row filter (55 % of the rows live, as the
bench's ragged lengths give): HIP-event time per launch and the implied HBM rate."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from item_alignment_amd import _lib
from item_alignment_amd._lib import check, stream_ptr
lib = _lib.load()
dev = torch.device("cuda:0")
B, L, H = 512, 255, 1024
M = B * L
g = torch.Generator(device="cpu").manual_seed(1)
lens = torch.randint(30, 255, (B,), generator=g)
live = (torch.arange(L)[None] < lens[:, None]).to(torch.uint8).to(dev).contiguous()
mk = live.view(M, 1).to(torch.bfloat16)
dy, dy2 = torch.randn(M, H, device=dev).to(torch.bfloat16) * mk, torch.randn(M, H, device=dev).to(torch.bfloat16) * mk
z = torch.randn(M, H, device=dev).to(torch.bfloat16)
mean, rstd = torch.zeros(M, device=dev), torch.ones(M, device=dev)
gamma = torch.ones(H, device=dev)
dz, dx = torch.empty_like(z), torch.empty_like(z)
dg, db, dbi = torch.zeros(H, device=dev), torch.zeros(H, device=dev), torch.zeros(H, device=dev)
wsb = lib.ia_ln_bwd_workspace_bytes(M, H)
ws = torch.empty(wsb, device=dev, dtype=torch.uint8)
outs = {}
for name, lp in (("all rows", None), ("row filter", live.data_ptr())):
    def run():
        check(lib.ia_ln_bwd2_rows(dy.data_ptr(), dy2.data_ptr(), None, z.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(), dz.data_ptr(),
                                  dx.data_ptr(), dg.data_ptr(), db.data_ptr(), dbi.data_ptr(), M, H, 0.1, 7, 3, lp, ws.data_ptr(), wsb, 0, stream_ptr()), "ln_bwd2_rows")
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1000 / 20
    outs[name] = (dz.clone(), dx.clone(), dg.clone())
    print(f"{name:12s}: {us:7.1f} us per launch (incl. the partial-sum reduce), live rows {live.float().mean().item():.2f}")
a, b = outs["all rows"], outs["row filter"]
print("identical:", all(torch.equal(x, y) for x, y in zip(a, b)))
