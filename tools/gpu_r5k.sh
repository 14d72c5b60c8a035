#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_setup_threads.py tests/test_gpu_adapter.py -m gpu -x -q > gpurun_out/r5k_tests.log 2>&1; tail -4 gpurun_out/r5k_tests.log
bash tools/gpu_r5f.sh
