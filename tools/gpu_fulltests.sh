#!/bin/bash
make -s -C tests/cpp/mini_g2o || exit 1
python -m pytest tests -x -q -m gpu > gpurun_out/full_tests.log 2>&1; tail -4 gpurun_out/full_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
