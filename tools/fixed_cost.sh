#!/bin/bash
# Fixed and per-step cost of bench.py's N > 1 timed region with ONE rank over real RCCL (run on the GPU box): ms per step at
# 20 / 100 / 400 timed steps for the sharded headline and for no exchange at all.   bash tools/fixed_cost.sh [extra bench args]
for n in 20 100 400; do
  for g in sharded none; do
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --force-gather --gather $g --steps $n --warmup 5 --no-gather-modes --no-gather-check --no-ftg "$@" 2>gpurun_out/fixed_cost.err | python -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print($n, '$g', round(d['ms_per_step'],4), round(d['ms_per_step']*$n,3), 'ms total')"
    grep 'timed\[' gpurun_out/fixed_cost.err | head -1
  done
done
