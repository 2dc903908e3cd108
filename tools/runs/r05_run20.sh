#!/bin/bash
python -m pytest tests/test_kernels_gpu.py -q -k "gemm" 2>&1 | grep -E "passed|failed|^E  |^FAILED" | cut -c1-300
cat > /tmp/ab_cmd.sh <<'EOS'
python tools/abl/dyn_ab.py 2>/dev/null | grep -E "static" | sed 's/| dynamic.*//' | head -8
python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-pmc --no-variants 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench:', round(d['value'],1), 'pairs/s', round(d['ms_per_step'],1), 'ms, frac', round(d['roofline']['frac'],4))"
EOS
bash tools/abl/lib_ab.sh tools/abl/lib_nopeel.so bash /tmp/ab_cmd.sh
