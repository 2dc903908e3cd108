#!/bin/bash
python -m pytest tests/test_kernels_gpu.py tests/test_models_gpu.py -q -x 2>&1 | grep -E "passed|failed|^E  |^FAILED" | cut -c1-300
cat > /tmp/ab_cmd.sh <<'EOS'
python tools/config_bench.py c3 c2 2>&1 | grep -E "pairs/s" | cut -c1-110
EOS
bash tools/abl/lib_ab.sh tools/abl/lib_nopeel.so bash /tmp/ab_cmd.sh
