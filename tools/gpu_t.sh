#!/bin/bash
# pytest on the GPU box with only the summary lines printed: bash tools/gpu_t.sh [test files / -k ...]
python -m pytest "${@:-tests}" -m gpu -x -q > gpurun_out/t.log 2>&1
grep -E "passed|failed|error|Error|assert" gpurun_out/t.log | tail -15
