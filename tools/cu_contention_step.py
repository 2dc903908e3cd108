#!/usr/bin/env python3
"""The bench's train step (CoCa roberta_large + ViT-B/16, fwd + bwd + fused AdamW) beside k whole CUs held by another stream's kernel
-- the step-level half of VERDICT r4 item 3 (tools/cu_contention.py is the per-GEMM half).  ia_debug_cu_hog(k, ms) is started on a
side stream and is resident before the timed steps begin; static tile order against the dynamic claim in ONE process
(ia_debug_gemm_dynamic), k = 0 / 8 / 16 / 32, at 16 and at 256 pairs per GPU.  The fair price of k missing CUs is 256 / (256 - k) on
the whole-chip kernels.

    python tools/cu_contention_step.py [pairs ...] > profiles/r05_cu_contention_step.txt
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import build_model, roberta_large_config
from item_alignment_amd import _lib
from item_alignment_amd.data.synthetic import SyntheticCocaPairs
from item_alignment_amd.models import functional as Fn

dev = torch.device("cuda:0")


def main():
    lib = _lib.load()
    side = torch.cuda.Stream()
    cfg = roberta_large_config()
    model = build_model(cfg, "vit_base_patch16_384", 2345).to(dev).train()
    arena = model.param_arena
    print("ms per train step, single stream, median of 3 blocks; x = against the same order's k = 0; fair price of k CUs = 256 / (256 - k)")
    for B in [int(a) for a in sys.argv[1:]] or [16, 256]:
        data = SyntheticCocaPairs(4 * B, image_size=cfg.image_size, seed=7)
        pool = [data.batch(list(range(g * B, (g + 1) * B)), dev, device_images=True) for g in range(2)]
        steps = 10 if B <= 32 else 1          # one hog launch lasts at most 1000 ms (ia_debug_cu_hog): the block has to fit under it

        def step(i):
            Fn.set_step_seed(1000 + i)
            arena.zero_grad()
            b = pool[i % 2]
            model(*b[:10], labels=b[10]).loss.backward()
            arena.adamw_step(1e-5, betas=(0.9, 0.98), eps=1e-8, weight_decay=1e-5)
        for i in range(3):
            step(i)
        torch.cuda.synchronize()
        est = None
        for mode in (0, 1):
            lib.ia_debug_gemm_dynamic(mode)
            row, base = [], None
            for hog in (0, 8, 16, 32):
                blocks = []
                for trial in range(3 if B <= 32 else 5):
                    torch.cuda.synchronize()
                    if hog:
                        # long enough to cover the block (a first estimate from the k = 0 block, x 2.2 for the slow-down, + margin)
                        _lib.check(lib.ia_debug_cu_hog(hog, float(min(1000.0, 2.2 * est * steps + 30.0)), side.cuda_stream), "ia_debug_cu_hog")
                        time.sleep(0.005)
                    t0 = time.perf_counter()
                    for i in range(steps):
                        step(i)
                    torch.cuda.current_stream().synchronize()
                    blocks.append((time.perf_counter() - t0) / steps * 1e3)
                    torch.cuda.synchronize()          # the hog runs out before the next block
                t = sorted(blocks)[len(blocks) // 2]
                if hog == 0:
                    base = t
                    est = t if est is None else est
                row.append(f"{t:7.1f} (x{t / base:5.3f})")
            print(f"{B:4d} pairs/step {'dynamic' if mode else 'static ':8s}| " + " | ".join(f"k={k:<2d} {r}" for k, r in zip((0, 8, 16, 32), row)), flush=True)
    lib.ia_debug_gemm_dynamic(0)


if __name__ == "__main__":
    main()
