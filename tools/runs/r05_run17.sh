#!/bin/bash
python -m pytest tests/test_models_gpu.py tests/test_baseline_shapes_gpu.py tests/test_dp_gpu.py -q -x 2>&1 | grep -E "passed|failed|^E  |^FAILED" | cut -c1-300
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-pmc --no-variants 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), 'pairs/s', round(d['ms_per_step'],1), 'ms', d['peak_hbm_gib'], 'GiB')"
