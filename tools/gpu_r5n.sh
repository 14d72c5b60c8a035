#!/bin/bash
python -m pytest tests/test_gpu_adapter.py -m gpu -x -q > gpurun_out/r5n_tests.log 2>&1; tail -3 gpurun_out/r5n_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
