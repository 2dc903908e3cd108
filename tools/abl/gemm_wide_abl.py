"""IA_GEMM_DBG ablations of one big GEMM (results are wrong with dbg bits set): python tools/abl/gemm_wide_abl.py [M N K aks bks]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from item_alignment_amd import ops
dev = torch.device("cuda:0")
M, N, K, aks, bks = (int(v) for v in sys.argv[1:6]) if len(sys.argv) > 5 else (8192, 8192, 8192, 0, 0)
a = torch.randn((K, M) if aks else (M, K), device=dev).bfloat16()
b = torch.randn((K, N) if bks else (N, K), device=dev).bfloat16()
f32 = bool(aks)
out = torch.empty((M, N), device=dev, dtype=torch.float32 if f32 else torch.bfloat16)
fn = lambda: ops.gemm(a, b, a_kstrided=bool(aks), b_kstrided=bool(bks), out=out, out_f32=f32)
for _ in range(3): fn()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(10): fn()
e.record(); torch.cuda.synchronize()
t = s.elapsed_time(e) / 10 * 1e-3
print(f"WIDE={os.environ.get('IA_GEMM_WIDE','0')} DBG={os.environ.get('IA_GEMM_DBG','0'):>4} M={M} N={N} K={K} {aks}{bks}: {t*1e6:8.1f} us {2*M*N*K/t/1e12:7.1f} TF/s")
