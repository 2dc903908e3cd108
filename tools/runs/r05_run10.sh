#!/bin/bash
python -m pytest tests/test_kernels_gpu.py -q -k "conv3x3" 2>&1 | grep "^E  \|^FAILED\|passed\|failed\|Error" | cut -c1-300 | head -20
python tools/conv_small_bench.py 32 2>&1 | tail -6
for n in 16 32; do IA_CB_PAIRS=$n python tools/config_bench.py c3 2>&1 | grep -E "pairs/s|HBM"; done
python -m pytest tests/test_models_gpu.py tests/test_baseline_shapes_gpu.py::test_c3_eca_nfnet_l0_at_800 -q -k "nfnet or resnet or image or c3" 2>&1 | grep "^E  \|^FAILED\|passed\|failed\|Error" | cut -c1-300 | head -20
bash tools/runs/r05_c3prof.sh 2>&1 | head -30
