G2OHIP_PLAN_DUMP=1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --dump-xp gpurun_out/xp_wv1.npy 2> gpurun_out/e1.log | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('wv=1', d['value'], round(d['kernels']['chol_factor(all levels)']['avg_ms'],4), d['residual_rel'])"
grep "group" gpurun_out/e1.log | head -3
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --opt wave_kernel=0 --dump-xp gpurun_out/xp_wv0.npy 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('wv=0', d['value'], round(d['kernels']['chol_factor(all levels)']['avg_ms'],4), d['residual_rel'])"
python -c "
import numpy as np
a=np.load('gpurun_out/xp_wv1.npy'); b=np.load('gpurun_out/xp_wv0.npy'); print('dx diff', np.abs(a-b).max()/np.abs(b).max())"
