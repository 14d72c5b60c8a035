#!/bin/bash
# round 5: the whole -m gpu suite, then the paths next to the headline number
python -m pytest tests -m gpu -x -q -s > gpurun_out/r5b_tests.log 2>&1; tail -4 gpurun_out/r5b_tests.log
bash tools/gpu_r5_paths.sh
