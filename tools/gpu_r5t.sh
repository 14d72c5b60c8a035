#!/bin/bash
# set-up split at the metric configuration through the plugin (G2OHIP_SETUP_TIMING / G2OHIP_ADAPTER_TIMING), then the whole GPU suite
B=tests/cpp/mini_g2o/build
for rep in 1 2; do
G2OHIP_SETUP_TIMING=1 G2OHIP_ADAPTER_TIMING=1 $B/g2o_host none $B/libg2o_solver_hip.so lm_fix6_3_hipdev 4 /tmp/su.json bench:100000:1000000:5:tight 2> gpurun_out/r5t_setup.err
cat gpurun_out/r5t_setup.err
python3 -c "
import json; d=json.load(open('/tmp/su.json')); print([round(i['iteration_s']*1e3,2) for i in d['iterations']], d['initializeOptimization_s'], d['graph_s'])"
done
python -m pytest tests -x -q -m gpu > gpurun_out/r5t_tests.log 2>&1; tail -5 gpurun_out/r5t_tests.log
