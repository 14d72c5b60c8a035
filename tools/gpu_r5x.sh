#!/bin/bash
make -s -C tests/cpp/mini_g2o || exit 1
python -m pytest tests/test_gpu_adapter.py tests/test_gpu_lm.py -x -q > gpurun_out/r5x_tests.log 2>&1; tail -5 gpurun_out/r5x_tests.log
