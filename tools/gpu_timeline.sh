G2OHIP_LIB=$PWD/variants/stamps/libg2ohip.so G2OHIP_CHOL_TIMELINE=$PWD/gpurun_out/timeline.txt python bench.py --steps 1 --warmup 1 --no-cpu-baseline --graph off > /dev/null 2>&1
python - <<EOP
import numpy as np
t=np.loadtxt("gpurun_out/timeline.txt")
s=t[:,1]-t[:,1].min(); e=t[:,2]-t[:,1].min()
print("launch span us", e.max()*0.01)
lv=[0,4096,6144,7168,7680,7936,8064,8128,8160,8176,8184,8188,8190,8191]
for i in range(13):
    a,b=lv[i],lv[i+1]
    print("level",i,"n",b-a,"start min/med/max %.1f %.1f %.1f"%(s[a:b].min()*0.01,np.median(s[a:b])*0.01,s[a:b].max()*0.01),"end min/med/max %.1f %.1f %.1f"%(e[a:b].min()*0.01,np.median(e[a:b])*0.01,e[a:b].max()*0.01),"dur med %.1f"%(np.median(e[a:b]-s[a:b])*0.01))
EOP
