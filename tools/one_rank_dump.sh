python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --force-gather --steps 20 --warmup 5 --no-gather-modes 2>gpurun_out/one_rank_dump.err > gpurun_out/one_rank_dump.json
python -c "
import json
d=json.loads([l for l in open('gpurun_out/one_rank_dump.json').read().splitlines() if l.startswith('{')][-1])
print(d['ms_per_step'], d['kernels_ms'], d['gather_modes']['sharded'].get('steady_state'), d.get('aborted'))
"
RC_BENCH_TRACE_TIMED=1 bash tools/fixed_cost.sh | grep -v none
