#!/bin/bash
mkdir -p gpurun_out/profiles
python tools/parity_report.py 2>&1 | grep -v "amdgpu.ids\|^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" > gpurun_out/profiles/r05_parity_report.txt; tail -14 gpurun_out/profiles/r05_parity_report.txt | cut -c1-200
python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|^E  |^FAILED" | cut -c1-300 > gpurun_out/r05_gpu_suite.txt; cat gpurun_out/r05_gpu_suite.txt
