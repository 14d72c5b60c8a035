for i in 1 2; do python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r3_b4_$i.log 2>&1; python - <<EOP
import json
d=json.loads([l for l in open("gpurun_out/r3_b4_$i.log") if l.startswith("{")][-1])
print(d["value"], {k:round(v["avg_ms"],4) for k,v in d["kernels"].items()}, d["roofline"]["kernel"])
EOP
done
NWS=1 bash tools/gpu_band_prof.sh 2>&1 | grep -E "band_wave|wave_front|schur_tile"
