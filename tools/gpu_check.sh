#!/bin/bash
# GPU parity suite, then the workload probe (each step under its own time limit)
set -e
timeout -k 10 400 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/gpu_tests.log 2>&1 || { tail -40 gpurun_out/gpu_tests.log; exit 1; }
tail -3 gpurun_out/gpu_tests.log
timeout -k 10 200 python -u tools/perf_probe.py "$@" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/probe.log
