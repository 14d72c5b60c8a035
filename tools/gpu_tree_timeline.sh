#!/bin/bash
# per-level start / end times of the tree levels (wave_front_kernel) of the metric configuration: stamps variant
# (make VARIANT=stamps EXTRA="-DG2OHIP_CHOL_STAMPS -DWSTAMP_BLOCK=2046"), once per option set given as arguments ("direct_children=0" ...)
for OPT in "$@"; do
G2OHIP_OPTIONS="$OPT" G2OHIP_LIB=$PWD/variants/stamps/libg2ohip.so G2OHIP_CHOL_STAMPS_PRINT=1 G2OHIP_CHOL_TIMELINE=$PWD/gpurun_out/timeline.txt python bench.py --steps 1 --warmup 1 --no-cpu-baseline --graph off 2> gpurun_out/stamps_err.txt > /dev/null
echo "== $OPT"
grep "^launch  0" gpurun_out/stamps_err.txt | tail -1
python - <<EOP
import numpy as np
t=np.loadtxt("gpurun_out/timeline.txt")
t=t[t[:,1]>0]
s=t[:,1]-t[:,1].min(); e=t[:,2]-t[:,1].min()
print("slots", len(t), "launch span us %.1f" % (e.max()*0.01))
a=0; n=(len(t)+1)//2
prev_end=0.0
while n>=1 and a<len(t):
    b=a+n
    print("level n %4d start min/med/max %6.1f %6.1f %6.1f  end min/med/max %6.1f %6.1f %6.1f  dur med %5.1f  last end - prev last end %5.1f"%(n,s[a:b].min()*0.01,np.median(s[a:b])*0.01,s[a:b].max()*0.01,e[a:b].min()*0.01,np.median(e[a:b])*0.01,e[a:b].max()*0.01,np.median(e[a:b]-s[a:b])*0.01, e[a:b].max()*0.01-prev_end))
    prev_end=e[a:b].max()*0.01
    a=b; n//=2
EOP
python tools/tree_handoff.py gpurun_out/timeline.txt
done
