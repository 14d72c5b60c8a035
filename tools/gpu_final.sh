#!/bin/bash
# end-of-round confirmation at HEAD: the whole GPU suite, smoke(), then every file profiles/ holds regenerated
python -m pytest tests -x -q -m gpu > gpurun_out/final_tests.log 2>&1; tail -3 gpurun_out/final_tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
bash tools/refresh_profiles.sh r5 > gpurun_out/final_refresh.log 2>&1; tail -3 gpurun_out/final_refresh.log
