#!/bin/bash
python -m pytest tests/test_models_gpu.py tests/test_kernels_gpu.py tests/test_baseline_shapes_gpu.py::test_c3_eca_nfnet_l0_at_800 -q -k "nfnet or resnet or image or c3 or conv or patch or nchw" 2>&1 | grep -E "passed|failed|^E  |^FAILED" | cut -c1-300
python tools/config_bench.py c3 c3r 2>&1 | grep -E "pairs/s"
