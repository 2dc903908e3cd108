import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from item_alignment_amd import _lib, ops
lib = _lib.load()
dev = torch.device("cuda:0")
for (M, N, K) in [(4352, 4096, 64), (4352, 4096, 256), (5000, 4096, 512), (65280, 1024, 1024)]:
    a = torch.randn((M, K), device=dev).to(torch.bfloat16); b = (torch.randn((N, K), device=dev) * 0.1).to(torch.bfloat16)
    lib.ia_debug_gemm_dynamic(0)
    ref = ops.gemm(a, b).clone(); torch.cuda.synchronize()
    print("static ok", M, N, K, flush=True)
    lib.ia_debug_gemm_dynamic(1)
    out = ops.gemm(a, b); torch.cuda.synchronize()
    print("dynamic ran", M, N, K, "equal:", torch.equal(out, ref), "max diff", (out.float() - ref.float()).abs().max().item(), flush=True)
    out = ops.gemm(a, b); torch.cuda.synchronize()
    print("dynamic 2nd ", torch.equal(out, ref), flush=True)
