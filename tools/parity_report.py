"""Runs the model-level parity tests on the GPU and prints what they measured against the reference golden vectors: relative
error of every output and cosine / relative error of every parameter gradient of every fixture (tests/test_models_gpu.py check()).
usage: python tools/parity_report.py > gpurun_out/parity.txt"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pytest  # noqa: E402

rc = pytest.main([os.path.join(ROOT, "tests", "test_models_gpu.py"), "-q", "-m", "gpu", "-p", "no:cacheprovider"])
T = sys.modules.get("test_models_gpu")
rows = sorted(set(T.MEASURED)) if T is not None else []
print(f"\npytest exit code {rc}; {len(rows)} measurements")
print(f"{'fixture':36s} {'kind':15s} {'tensor':62s} value")
for case, kind, key, val in rows:
    print(f"{case:36s} {kind:15s} {key:62s} {val:.4f}")
cos = [v for _, k, _, v in rows if k == "grad cos"]
if cos:
    print(f"\ngradient cosine: min {min(cos):.4f}, median {sorted(cos)[len(cos) // 2]:.4f} over {len(cos)} tensors")
gr = {(c, t): v for c, k, t, v in rows if k == "grad rel"}
nz = {(c, t): v for c, k, t, v in rows if k == "grad rel-bf16-oracle"}
if gr:
    worst = sorted(gr.items(), key=lambda kv: -kv[1])[:8]
    print(f"gradient magnitude error (of the reference's max): max {max(gr.values()):.4f}, median {sorted(gr.values())[len(gr) // 2]:.4f}; "
          "largest, with the distance to the bf16-rounding oracle's gradient where the fp32 bar (5e-2) failed:")
    for (c, t), v in worst:
        print(f"    {c:34s} {t:58s} {v:.4f}   vs bf16 oracle {nz.get((c, t), float('nan')):.4f}")
rels = [v for _, k, _, v in rows if k == "out rel"]
if rels:
    print(f"output relative error: max {max(rels):.4f}, median {sorted(rels)[len(rels) // 2]:.4f} over {len(rels)} tensors")
