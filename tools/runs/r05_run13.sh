#!/bin/bash
mkdir -p gpurun_out/profiles
python -m pytest tests/test_kernels_gpu.py tests/test_optim_gpu.py -q -k "attention or attn or optim" 2>&1 | grep -E "passed|failed|^E  |^FAILED" | cut -c1-300
python tools/abl/fwd_split_ab.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/profiles/r05_ab_attention_fwd_split.txt
for s in 1 0 1 0; do
  IA_ATTN_FWD_SPLIT=$s python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-pmc --no-variants 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('IA_ATTN_FWD_SPLIT=$s:', round(d['value'],1), 'pairs/s', round(d['ms_per_step'],1), 'ms')"
done | tee -a gpurun_out/profiles/r05_ab_attention_fwd_split.txt
