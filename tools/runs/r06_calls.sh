#!/bin/bash
# Every GPU command list of round 6 other than the final collection (r06_final.sh), in the order they ran: `r06_calls.sh <name>` runs one
# (through gpu.sh -> gpurun, from the repository root on the box).  One file instead of one script per call (round 5 left 21 of those).
cd ${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
case "$1" in
  baseline)
    bash tools/runs/run.sh suite
    bash tools/runs/run.sh bench
    ;;
  call2)
    bash tools/runs/run.sh tests tests/test_models_gpu.py -k dropout_on -s
    bash tools/runs/run.sh suite
    ;;
  call3)
    IA_DROPOUT_SEEDS=256 python -m pytest tests/test_models_gpu.py -k dropout_on -s -q 2>&1 | grep -E "dropout z|passed|failed"
    IA_DROPOUT_SEEDS=64 python -m pytest tests/test_models_gpu.py -k dropout_on -s -q 2>&1 | grep -E "dropout z|passed|failed"
    python tools/c5x_mem_probe.py 16 2>&1 | grep -v "Warning\|amdgpu.ids\|getattr" | tee gpurun_out/r06_c5x_mem_probe16.txt
    PROF_STEPS=11 bash tools/runs/run.sh prof c5x python3 tools/config_bench.py c5x
    ;;
  call4)
    O=gpurun_out/r06_call4.txt; : > $O
    python -m pytest tests/test_models_gpu.py -k "coca" -q 2>&1 | tail -3 >> $O
    for n in 16 32 64 96; do IA_CB_PAIRS=$n python tools/config_bench.py c5x 2>&1 | grep -E "pairs/s|Error|error" >> $O; done
    IA_DROPOUT_SEEDS=256 python -m pytest tests/test_models_gpu.py -k dropout_on -s -q 2>&1 | grep -E "dropout z|passed|failed" >> $O
    IA_DROPOUT_SEEDS=64 python -m pytest tests/test_models_gpu.py -k dropout_on -s -q 2>&1 | grep -E "dropout z|passed|failed" >> $O
    cat $O
    ;;
  c5x)
    # C5x (--ensemble cross_attn, coca_large): where the memory goes + a kernel profile at the round-5 batch (verdict r5 item 4)
    python tools/c5x_mem_probe.py 4 2>&1 | grep -v Warning | tee gpurun_out/r06_c5x_mem_probe.txt
    PROF_STEPS=11 bash tools/runs/run.sh prof c5x python3 tools/config_bench.py c5x
    ;;
  call5)
    O=gpurun_out/r06_call5.txt; : > $O
    python -m pytest tests/test_kernels_gpu.py -k "eca or conv3x3" -q -x 2>&1 | tail -4 >> $O
    python -m pytest tests/test_models_gpu.py -k "nfnet or dropout_on or resnet" -q -x 2>&1 | tail -4 >> $O
    python -m pytest tests/test_baseline_shapes_gpu.py -k "c3 or c5" -q -x 2>&1 | tail -4 >> $O
    python -m pytest tests/test_optim_gpu.py -q -x 2>&1 | tail -3 >> $O
    for v in 1 0 1 0; do echo "IA_ECA_LINEAR=$v" >> $O; IA_ECA_LINEAR=$v python tools/config_bench.py c3 2>&1 | grep -E "pairs/s" >> $O; done
    PROF_STEPS=11 bash tools/runs/run.sh prof c5x64 python3 tools/config_bench.py c5x > /dev/null 2>&1
    PROF_STEPS=11 bash tools/runs/run.sh prof c3 python3 tools/config_bench.py c3 > /dev/null 2>&1
    cat $O
    ;;
  call6)
    O=gpurun_out/r06_call6.txt; : > $O
    python -m pytest tests/test_kernels_gpu.py -k "exact_delta" -q -x -s 2>&1 | grep -E "adverse|passed|failed|Error|assert" | cut -c1-400 >> $O
    python -m pytest tests -m gpu -x -q 2>&1 | tail -12 >> $O
    cat $O
    ;;
  call7)
    O=gpurun_out/r06_call7.txt; : > $O
    python -m pytest tests -m gpu -x -q 2>&1 | tail -12 >> $O
    for v in 1 0 1 0; do echo "IA_NFNET_FUSE_TAIL=$v (32 pairs)" >> $O; IA_CB_PAIRS=32 IA_NFNET_FUSE_TAIL=$v python tools/config_bench.py c3 2>&1 | grep -E "pairs/s" >> $O; done
    python tools/config_bench.py c3 2>&1 | grep -E "pairs/s" >> $O
    bash tools/runs/run.sh quick >> $O 2>&1
    cat $O
    ;;
  call8)
    O=gpurun_out/r06_call8.txt; : > $O
    for i in 1 2 3; do python -m pytest tests/test_optim_gpu.py -q -x 2>&1 | grep -E "^E  .*(AssertionError|assert|\(')|passed|failed" | cut -c1-400 | head -8 >> $O; done
    cat $O
    ;;
  call9)
    bash tools/runs/run.sh suite
    O=gpurun_out/r06_ab_fused_exchange.txt; : > $O
    echo "attn_bwd_fused_kernel<true> at 512 x 255 x 16, dropout 0.1 (tools/attn_one.py under rocprofv3): this commit's exchange layout (dense rows + XOR key)" >> $O
    echo "against the previous one (72-byte rows), two libraries on one box, interleaved" >> $O
    cp item_alignment_amd/libitemalign_hip.so /tmp/lib_default.so
    for rep in 1 2; do
      for which in default prev; do
        if [ $which = prev ]; then cp tools/abl/lib_prev_exchange.so item_alignment_amd/libitemalign_hip.so; else cp /tmp/lib_default.so item_alignment_amd/libitemalign_hip.so; fi
        TAG=r06tmp bash tools/runs/run.sh prof bwdf_$which python3 tools/attn_one.py 512 255 16 0.1 > /dev/null 2>&1
        echo "$which (rep $rep): $(grep -E 'attn_bwd_fused' gpurun_out/r06tmp_bwdf_${which}_kernel_stats_summary.txt)" >> $O
      done
    done
    cp /tmp/lib_default.so item_alignment_amd/libitemalign_hip.so
    cat $O
    bash tools/runs/run.sh quick
    ;;
  call11)
    O=gpurun_out/r06_call11.txt; : > $O
    python -m pytest tests/test_kernels_gpu.py -k "skips_query or eca_block or attention_fwd_bwd or bwd_bias or dropout_mask or bench_shapes or first_launch" -q -x --tb=short 2>&1 | grep -E "^E  |passed|failed" | cut -c1-400 | tail -6 >> $O
    python -m pytest tests/test_models_gpu.py tests/test_engine_gpu.py -q -x --tb=short 2>&1 | grep -E "^E  |passed|failed|^FAILED" | cut -c1-400 | tail -6 >> $O
    python -m pytest tests/test_baseline_shapes_gpu.py -q -x --tb=short 2>&1 | grep -E "^E  |passed|failed|^FAILED" | cut -c1-400 | tail -6 >> $O
    for v in 1 0 1 0; do echo "IA_MASKED_ROWS_DEAD=$v" >> $O; IA_MASKED_ROWS_DEAD=$v bash tools/runs/run.sh quick >> $O 2>&1; done
    for v in 1 2 1 2; do echo "IA_NFNET_FUSE_TAIL=$v (32 pairs)" >> $O; IA_CB_PAIRS=32 IA_NFNET_FUSE_TAIL=$v python tools/config_bench.py c3 2>&1 | grep -E "pairs/s" >> $O; done
    cat $O
    ;;
  call12)
    O=gpurun_out/r06_call12.txt; : > $O
    python -m pytest tests/test_models_gpu.py tests/test_engine_gpu.py tests/test_cli_gpu.py tests/test_dp_gpu.py tests/test_optim_gpu.py -q -x --tb=short 2>&1 | grep -E "^E  |passed|failed|^FAILED" | cut -c1-400 | tail -8 >> $O
    cat $O
    ;;
  call13)
    O=gpurun_out/r06_call13.txt; : > $O
    python -m pytest tests/test_kernels_gpu.py -k "skips_query or attention" -q -x --tb=short 2>&1 | grep -E "^E  |passed|failed|^FAILED" | cut -c1-400 | tail -6 >> $O
    bash tools/runs/run.sh suite >> $O 2>&1
    for v in 1 0 1 0; do echo "IA_MASKED_ROWS_DEAD=$v" >> $O; IA_MASKED_ROWS_DEAD=$v python tools/config_bench.py c2 2>&1 | grep -E "pairs/s" >> $O; done
    for v in 1 0; do echo "IA_MASKED_ROWS_DEAD=$v" >> $O; IA_MASKED_ROWS_DEAD=$v bash tools/runs/run.sh quick >> $O 2>&1; done
    cat $O
    ;;
  call14)
    O=gpurun_out/r06_call14.txt; : > $O
    python -m pytest tests/test_engine_gpu.py tests/test_models_gpu.py -q -x --tb=short 2>&1 | grep -E "^E  |passed|failed|^FAILED" | cut -c1-400 | tail -6 >> $O
    for v in 1 0 1 0; do echo "IA_LN_ROWS=$v: $(IA_LN_ROWS=$v bash tools/runs/run.sh quick 2>&1 | tail -1)" >> $O; done
    cat $O
    ;;
  call15)
    bash tools/runs/run.sh suite
    ;;
  call17)
    O=gpurun_out/r06_call17.txt; : > $O
    python tools/abl/ln_rows_bench.py >> $O 2>&1
    python -m pytest tests/test_models_gpu.py -q -x --tb=short -k "pkgm" 2>&1 | grep -E "^E  |passed|failed|^FAILED" | cut -c1-300 | tail -4 >> $O
    for v in 1 0; do
      IA_LN_ROWS=$v TAG=r06tmp bash tools/runs/run.sh prof lnrows$v python3 bench.py --single-stream --steps 3 --warmup 2 --no-cpu-baseline --no-pmc --no-variants > /dev/null 2>&1
      echo "IA_LN_ROWS=$v: $(grep -E 'ln_bwd_kernel' gpurun_out/r06tmp_lnrows${v}_kernel_stats_summary.txt)" >> $O
    done
    for v in 1 0; do echo "IA_MASKED_ROWS_DEAD=$v" >> $O; IA_MASKED_ROWS_DEAD=$v python tools/config_bench.py c4 2>&1 | grep -E "pairs/s" >> $O; done
    cat $O
    ;;
  call18)
    O=gpurun_out/r06_call18.txt; : > $O
    python tools/abl/ln_rows_bench.py 2>&1 | grep -v amdgpu.ids >> $O
    python -m pytest tests/test_engine_gpu.py tests/test_models_gpu.py tests/test_kernels_gpu.py -q -x --tb=short 2>&1 | grep -E "^E  |passed|failed|^FAILED" | cut -c1-300 | tail -4 >> $O
    for v in 1 0; do
      IA_LN_ROWS=$v TAG=r06tmp bash tools/runs/run.sh prof lnrows$v python3 bench.py --single-stream --steps 3 --warmup 2 --no-cpu-baseline --no-pmc --no-variants > /dev/null 2>&1
      echo "IA_LN_ROWS=$v: $(grep -E 'ln_bwd_kernel' gpurun_out/r06tmp_lnrows${v}_kernel_stats_summary.txt)" >> $O
    done
    for v in 1 0 1 0; do echo "IA_LN_ROWS=$v: $(IA_LN_ROWS=$v bash tools/runs/run.sh quick 2>&1 | tail -1)" >> $O; done
    cat $O
    ;;
  call20)
    O=gpurun_out/r06_call20.txt; : > $O
    timeout 600 python -m pytest tests/test_kernels_gpu.py -k "lookahead or gemm" -q -x --tb=short 2>&1 | grep -E "^E  |passed|failed|^FAILED" | cut -c1-300 | tail -6 >> $O
    timeout 600 python tools/abl/gemm_la_ab.py 2>&1 | grep -v amdgpu.ids >> $O
    for v in 1 0 1 0; do echo "IA_GEMM_LA=$v: $(IA_GEMM_LA=$v timeout 600 bash tools/runs/run.sh quick 2>&1 | tail -1)" >> $O; done
    cat $O
    ;;
  final)        # GroupNorm kernel tests on the final library, then the round's collection (tools/runs/r06_final.sh)
    O=gpurun_out/r06_final_head.txt; : > $O
    timeout 900 python -m pytest tests/test_kernels_gpu.py -k "groupnorm or ring_of_zeros or batchnorm" -q -x --tb=short 2>&1 | grep -E "^E  |passed|failed|^FAILED" | cut -c1-400 | tail -8 >> $O
    cat $O
    bash tools/runs/r06_final.sh
    ;;
  s2)           # the strided 3x3 convolutions without a patch matrix: kernel tests, the NF-Net model tests, C3 A/B on one box
    O=gpurun_out/r06_s2.txt; : > $O
    timeout 900 python -m pytest tests/test_kernels_gpu.py -k "stride2 or conv" -q -x --tb=short 2>&1 | grep -E "^E  |passed|failed|^FAILED" | cut -c1-400 | tail -8 >> $O
    timeout 900 python -m pytest tests/test_models_gpu.py tests/test_cli_gpu.py -k "nfnet or image" -q --tb=short 2>&1 | grep -E "^E  |passed|failed|^FAILED" | cut -c1-400 | tail -8 >> $O
    for v in 1 0 1 0; do echo "IA_CONV_S2_DIRECT=$v: $(IA_CONV_S2_DIRECT=$v timeout 600 python tools/config_bench.py c3 2>&1 | grep -E 'pairs/s')" >> $O; done
    cat $O
    ;;
  c3prof)       # C3 again after the strided kernels: kernel summary, PMC traffic, A/B lines (-> profiles/r06_c3_*.txt, r06_ab_c3_strided.txt)
    mkdir -p gpurun_out/profiles; OUT=$GRAFT_REPO_ROOT/gpurun_out/profiles; R=$GRAFT_REPO_ROOT
    O=$OUT/r06_ab_c3_strided.txt; : > $O
    for v in "1 2" "1 0" "0 0" "1 2" "1 0" "0 0"; do set -- $v
      echo "IA_CONV_S2_DIRECT=$1 IA_CONV_S2_DGRAD=$2: $(IA_CONV_S2_DIRECT=$1 IA_CONV_S2_DGRAD=$2 timeout 600 python tools/config_bench.py c3 2>&1 | grep -E 'pairs/s')" >> $O; done
    cat $O
    timeout 900 python - >> $O 2>&1 <<'PY'
# soak: 200 C3 steps on fresh random images, the loss stays finite and falls
import sys, torch
sys.path.insert(0, ".")
import item_alignment_amd.models as M
from item_alignment_amd.optim import AdamW
from types import SimpleNamespace
dev = torch.device("cuda:0")
cfg = SimpleNamespace(num_labels=2, hidden_dropout_prob=0.1, loss_type="ce", loss_margin=0.0, classification_method="cls", hidden_size=2304)
torch.manual_seed(1)
m = M.NFNetTwoTower(cfg, M.create_model("eca_nfnet_l0")).to(dev).train()
opt = AdamW(m.parameters(), lr=2e-5)
g = torch.Generator().manual_seed(0)
im1, im2 = torch.randn((8, 3, 413, 397), generator=g).to(dev), torch.randn((8, 3, 413, 397), generator=g).to(dev)      # odd, ragged maps
labels = torch.randint(0, 2, (8,), generator=g).to(dev)
losses = []
for step in range(200):
    out = m(im1, im2, labels)
    out.loss.backward()
    opt.step(); opt.zero_grad()
    losses.append(float(out.loss))
assert all(l == l and abs(l) < 1e4 for l in losses), losses[-5:]
print(f"soak: 200 steps of eca_nfnet_l0 two_tower at 413 x 397 (odd maps through every strided kernel): loss {losses[0]:.4f} -> {losses[-1]:.4f}, all finite")
PY
    tail -1 $O
    (cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $OUT/kt_c3 -o b --output-format csv -- python3 $R/tools/config_bench.py c3 > $OUT/c3_log.txt 2>&1)
    python3 tools/config_bench.py --pmc c3 2>&1 | grep -E "pairs/s|HBM traffic" > $OUT/r06_c3_hbm.txt
    python3 tools/prof_summary.py $OUT/kt_c3/b_kernel_stats.csv 11 40 > $OUT/r06_c3_kernel_stats_summary.txt
    rm -rf $OUT/kt_c3 $OUT/c3_log.txt
    cat $OUT/r06_c3_hbm.txt; head -12 $OUT/r06_c3_kernel_stats_summary.txt
    ;;
  s2d)          # the sliced data gradient of the stem's strided convolution: tests, then IA_CONV_S2_DGRAD=2 (with it) / 1 (without) on one box
    O=gpurun_out/r06_s2d.txt; : > $O
    timeout 900 python -m pytest tests/test_kernels_gpu.py -k "stride2" -q -x --tb=short 2>&1 | grep -E "^E  |passed|failed|^FAILED" | cut -c1-400 | tail -6 >> $O
    timeout 900 python -m pytest tests/test_models_gpu.py tests/test_cli_gpu.py -k "nfnet or image" -q --tb=short 2>&1 | grep -E "^E  |passed|failed|^FAILED" | cut -c1-400 | tail -6 >> $O
    for m in 2 1 2 1; do echo "IA_CONV_S2_DGRAD=$m: $(IA_CONV_S2_DGRAD=$m timeout 600 python tools/config_bench.py c3 2>&1 | grep -E 'pairs/s')" >> $O; done
    cat $O
    ;;
  s2w)          # the one-launch weight gradient of the strided convolutions against the four masked launches
    O=gpurun_out/r06_s2w.txt; : > $O
    for m in 1 0; do
      echo "IA_CONV_S2_WGRAD_MERGED=$m: $(IA_CONV_S2_WGRAD_MERGED=$m timeout 900 python -m pytest tests/test_kernels_gpu.py -k 'stride2' -q -x --tb=short 2>&1 | grep -E '^E  |passed|failed|^FAILED' | cut -c1-300 | tail -3)" >> $O
    done
    timeout 900 python -m pytest tests/test_models_gpu.py -k "nfnet" -q --tb=short 2>&1 | grep -E "^E  |passed|failed|^FAILED" | cut -c1-400 | tail -4 >> $O
    for m in 1 0 1 0; do echo "IA_CONV_S2_WGRAD_MERGED=$m: $(IA_CONV_S2_WGRAD_MERGED=$m timeout 600 python tools/config_bench.py c3 2>&1 | grep -E 'pairs/s')" >> $O; done
    cat $O
    ;;
  bit)          # the BiT towers: kernel + model + CLI tests, parity numbers, throughput beside resnetv2_50
    O=gpurun_out/r06_bit.txt; : > $O
    timeout 900 python -m pytest tests/test_kernels_gpu.py -k "groupnorm or ring_of_zeros or maxpool or batchnorm" -q -x --tb=short 2>&1 | grep -E "^E  |passed|failed|^FAILED" | cut -c1-400 | tail -8 >> $O
    timeout 900 python -m pytest tests/test_cli_gpu.py -k "finetune_image" -q -x --tb=short 2>&1 | grep -E "^E  |passed|failed|^FAILED" | cut -c1-400 | tail -8 >> $O
    timeout 1500 python tools/parity_report.py > gpurun_out/r06_parity_report.txt 2>&1
    grep -E "^E  |passed|failed|^FAILED|bit|pytest exit" gpurun_out/r06_parity_report.txt | cut -c1-400 | tail -60 >> $O
    timeout 900 python tools/config_bench.py c3r c3b 2>&1 | grep -E "pairs/s|GiB|Error|error" >> $O
    timeout 900 python tools/config_bench.py c3b3 2>&1 | grep -E "pairs/s|GiB|Error|error" >> $O
    cat $O
    ;;
  *) echo "usage: $0 <" $(grep -oE '^  [a-z0-9_]+\)' "$0" | tr -d ' )') ">" >&2; exit 2 ;;
esac
