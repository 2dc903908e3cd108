#!/usr/bin/env python3
"""How much do the large GEMMs slow down when another stream's kernel holds k whole CUs?

The stand-in for an RCCL all-reduce that overlaps the backward GEMMs: ia_debug_cu_hog(k, ms) -- k workgroups x 160 KiB of LDS, so
nothing co-resides on their CUs -- is started on a side stream; once it is resident the GEMM is launched `reps` times on the main stream
and timed with events.  The fair price of k missing CUs is 256 / (256 - k); the static tile order (IA_GEMM_DYNAMIC=0) pays ~2 x for any
k > 0 (the workgroups that cannot start keep their tiles until a sibling has finished all of its own).

Both tile orders are measured in ONE process on one box (ia_debug_gemm_dynamic), k = 0 / 8 / 16 / 32 hogged CUs each.  Note on the
k > 0 columns: the hog sleeps, so the chip's power budget is shared by fewer busy CUs and the GEMM's clock RISES -- a dynamic launch
beside 8-32 sleeping CUs can come out faster than alone; the comparison that matters is static against dynamic at the same k.

    python tools/cu_contention.py > profiles/r05_cu_contention.txt
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from item_alignment_amd import _lib, ops

dev = torch.device("cuda:0")


def main():
    lib = _lib.load()
    side = torch.cuda.Stream()
    M = 65280                      # 256 sequences x 255 tokens: the bench's text-tower rows
    shapes = [("qkv     NT bias      ", M, 3072, 1024, 0, 0, ops.EPI_BIAS), ("ffn1    NT bias+gelu ", M, 4096, 1024, 0, 0, ops.EPI_BIAS_GELU),
              ("ffn2    NT plain     ", M, 1024, 4096, 0, 0, ops.EPI_NONE), ("dgrad   NN plain     ", M, 4096, 1024, 0, 1, ops.EPI_NONE),
              ("dgrad   NN x gelu'   ", M, 4096, 1024, 0, 1, ops.EPI_DGELU), ("wgrad   TN split-K   ", 4096, 1024, M, 1, 1, ops.EPI_NONE),
              ("small M NT (16 pairs)", 8160, 4096, 1024, 0, 0, ops.EPI_BIAS)]
    reps = 4
    print(f"times in us per launch ({reps} launches under one 30-ms hog, median of 3); x = against the same order's k = 0; fair price of k CUs = 256 / (256 - k)")
    print(f"{'gemm':22s} {'M':>6s} {'N':>5s} {'K':>6s} {'order':8s}| " + " | ".join(f"k={k:<3d} (fair x{256 / (256 - k):.3f})" for k in (0, 8, 16, 32)))
    for name, m, n, k_, aks, bks, epi in shapes:
        a = torch.randn((k_, m) if aks else (m, k_), device=dev).to(torch.bfloat16)
        b = (torch.randn((k_, n) if bks else (n, k_), device=dev) * 0.05).to(torch.bfloat16)
        f32 = bool(aks)
        out = torch.empty((m, n), device=dev, dtype=torch.float32 if f32 else torch.bfloat16)
        kw = {}
        if epi in (ops.EPI_BIAS, ops.EPI_BIAS_GELU):
            kw["bias"] = torch.zeros(n, device=dev)
        if epi == ops.EPI_BIAS_GELU:
            kw["pre_out"] = torch.empty_like(out)
        if epi == ops.EPI_DGELU:
            kw["aux"] = torch.randn((m, n), device=dev).to(torch.bfloat16)

        def run():
            ops.gemm(a, b, a_kstrided=bool(aks), b_kstrided=bool(bks), out=out, out_f32=f32, epilogue=epi, **kw)
        for _ in range(5):
            run()
        torch.cuda.synchronize()
        for mode in (0, 1):
            lib.ia_debug_gemm_dynamic(mode)
            row = []
            base = None
            for hog in (0, 8, 16, 32):
                best = []
                for trial in range(3):
                    torch.cuda.synchronize()
                    if hog:
                        _lib.check(lib.ia_debug_cu_hog(hog, 30.0, side.cuda_stream), "ia_debug_cu_hog")
                        time.sleep(0.003)                       # the hog is resident before the first GEMM workgroup is dispatched
                    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    s.record()
                    for _ in range(reps):
                        run()
                    e.record()
                    torch.cuda.synchronize()
                    best.append(s.elapsed_time(e) / reps * 1e3)
                t = sorted(best)[1]
                base = t if hog == 0 else base
                row.append(f"{t:8.1f} (x{t / base:5.3f})   ")
            print(f"{name:22s} {m:6d} {n:5d} {k_:6d} {'dynamic' if mode else 'static':8s}| " + " | ".join(row))


if __name__ == "__main__":
    main()
