#!/bin/bash
# profiles/r5_paths.jsonl: the paths next to the headline number (run on the GPU box from the repo root)
OUT=gpurun_out/r5_paths.jsonl
: > $OUT
python tools/paths_bench.py >> $OUT 2> gpurun_out/r5_paths.err
for variant in "--edge-data arrays --no-cpu-baseline" "--information edge" "--edge-data arrays --information edge --no-cpu-baseline" ""; do
  python bench.py --steps 20 --warmup 5 $variant 2>> gpurun_out/r5_paths.err | python -c "
import json, sys
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
row = {'path': 'config 4 (metric configuration), bench.py $variant'.rstrip(), 'value_ms_per_iter': d['value'], 'edge_data': d['config']['edge_data'],
       'information': d['config']['information'], 'roofline': d['roofline'],
       'kernels_ms': {k: round(v['avg_ms'] * v['launches_per_step'], 4) for k, v in d['kernels'].items()},
       'cpu_baseline': {k: d['cpu_baseline'][k] for k in ('value', 'unit', 'cores', 'kind', 'breakdown_ms', 'host')} if 'cpu_baseline' in d else None,
       'dx_rel_err': d.get('dx_rel_err'), 'chi2_rel_err': d.get('chi2_rel_err'), 'residual_rel': d.get('residual_rel')}
print(json.dumps(row))" >> $OUT
done
cat $OUT | cut -c1-400
