import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from item_alignment_amd import ops
dev = torch.device("cuda:0")
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
for H in (512, 768, 1024, 1536, 2048):
    M = (147712 * 768 // H) // 4 * 4
    x = torch.randn((M, H), device=dev).bfloat16(); r = torch.randn((M, H), device=dev).bfloat16()
    g = torch.ones(H, device=dev); b = torch.zeros(H, device=dev)
    t = timeit(lambda: ops.ln_fwd(x, g, b, 1e-5, residual=r))
    print(f"ln fwd M={M} H={H}: {t*1e6:7.1f} us {M*H*2*4/t/1e9:7.0f} GB/s")
    y, z, mean, rstd = ops.ln_fwd(x, g, b, 1e-5, residual=r)
    dy = torch.randn((M, H), device=dev).bfloat16()
    t = timeit(lambda: ops.ln_bwd(dy, z, mean, rstd, g))
    print(f"ln bwd M={M} H={H}: {t*1e6:7.1f} us {M*H*2*3/t/1e9:7.0f} GB/s")
