#!/bin/bash
# C5x (--ensemble cross_attn, coca_large): where the memory goes + a kernel profile at the round-5 batch (verdict r5 item 4)
cd $GRAFT_REPO_ROOT
python tools/c5x_mem_probe.py 4 2>&1 | grep -v Warning | tee gpurun_out/r06_c5x_mem_probe.txt
PROF_STEPS=11 bash tools/runs/run.sh prof c5x python3 tools/config_bench.py c5x
