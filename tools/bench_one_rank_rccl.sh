#!/bin/bash
# The N > 1 code path of bench.py with ONE rank and the real RCCL backend (run on the GPU box): the in-place
# `full-u16` all-gather through torch.distributed, then the same through the C-ABI's rc_gather_trajectory.
for via in torch abi; do
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py \
      --gpus 1 --force-gather --steps 60 --warmup 10 --gather-via $via 2>gpurun_out/one_rank_$via.err | \
    python -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('$via', round(d['ms_per_step'],4), d['config']['gather'], {k:(round(v['ms_per_step'],4), v['bytes_per_gpu_per_step']) for k,v in d['gather_modes'].items()})"
done
