#!/bin/bash
# round-4 measurement pass: big loops parity case, loop-closure graphs at scale, config-5 LM kernel trace
python -m pytest "tests/test_gpu_lm.py::test_ba_graph_with_loop_closures_and_ragged_lists" -m gpu -x -q 2>&1 | tail -3
: > gpurun_out/r4_loops_scale.jsonl
python tools/probe/loops_scale.py 10000 100000 5 >> gpurun_out/r4_loops_scale.jsonl 2> gpurun_out/loops.err
HUBS=1 python tools/probe/loops_scale.py 4500 20000 5 >> gpurun_out/r4_loops_scale.jsonl 2>> gpurun_out/loops.err
HUBS=1 python tools/probe/loops_scale.py 10000 60000 5 >> gpurun_out/r4_loops_scale.jsonl 2>> gpurun_out/loops.err
cut -c1-330 gpurun_out/r4_loops_scale.jsonl
bash tools/gpu_lm_trace.sh; tail -2 gpurun_out/lmtrace/run.log | cut -c1-300; tail -60 gpurun_out/lmtrace/seq.txt
