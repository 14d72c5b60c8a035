#!/bin/bash
# Gauss-Newton (no damping) through the plugin at several sizes: which iteration fails, if any
B=tests/cpp/mini_g2o/build
for sz in 2000:20000 20000:200000 100000:1000000; do
for solver in gn_fix6_3_hipdev gn_fix6_3_hip; do
  timeout 900 $B/g2o_host none $B/libg2o_solver_hip.so $solver 4 /tmp/gn.json bench:$sz:5:tight 2> /tmp/gn.err; rc=$?
  echo "$solver $sz rc $rc: $(grep -i "fail\|not pos" /tmp/gn.err | head -2 | tr '\n' ' ')"
  [ $rc = 0 ] && python3 -c "
import json; d=json.load(open('/tmp/gn.json')); print('   chi2_initial', d['chi2_initial'], 'iterations ms', [round(i['iteration_s']*1e3,2) for i in d['iterations']], 'chi2', d['iterations'][-1]['chi2'])"
done; done 2>&1 | tee gpurun_out/r5u_gn.txt
