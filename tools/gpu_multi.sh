# functional check of the N>1 bench paths on a 1-GPU box (staged exchange: timing meaningless)
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 3 --warmup 1 --comm staged --poses 20000 --landmarks 200000 --no-cpu-baseline 2>&1 | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('bench N=2 staged:', d['value'], d['solve_ok'], d.get('collectives'), d['config']['parallelism'])"
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29513 lm_bench.py --gpus 2 --comm staged --poses 20000 --landmarks 200000 --iterations 4 2>&1 | grep "^{" | cut -c1-700
python lm_bench.py --poses 20000 --landmarks 200000 --iterations 4 | cut -c1-600
python bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('bench N=1:', d['value'], d['roofline']['kernel'], round(d['roofline']['avg_launch_ms'],4))"
