python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -5
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/b3.log 2>&1; python - <<EOP
import json
d=json.loads([l for l in open("gpurun_out/b3.log") if l.startswith("{")][-1])
print(d["value"], {k:round(v["avg_ms"],4) for k,v in d["kernels"].items()}, d["residual_rel"])
EOP
G2OHIP_LIB=$PWD/variants/stamps/libg2ohip.so G2OHIP_CHOL_STAMPS_PRINT=1 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --graph off 2>&1 | grep "^launch  0" | tail -1
