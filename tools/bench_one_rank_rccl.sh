#!/bin/bash
# The N > 1 code path of bench.py with ONE rank and the real RCCL backend (run on the GPU box), the driver's own step counts: the
# sharded headline (ring per rank + summary per step + a packed 50 x 50 batch every 10th step, on a side stream), its steady state,
# and every per-step record gather through torch.distributed, the C-ABI's rc_gather_trajectory and the peer-copy transport (with
# one rank: the local copy and the flag protocol without peers), each with its self-check.
mkdir -p gpurun_out
for via in torch abi p2p; do
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py \
      --gpus 1 --force-gather --steps 20 --warmup 5 --gather-via $via 2>gpurun_out/one_rank_$via.err | \
    python -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); gm=d['gather_modes']; print('$via', 'headline', d['config']['gather'], round(d['ms_per_step'],4), 'ms/step; steady', round(gm['sharded']['steady_state']['ms_per_step'],4), 'rccl_ranks', d['config']['rccl_ranks'], 'abi', d['config']['abi_comm_ranks'], 'gather_check', d['gather_check']['ok'], sorted(d['gather_check']['payloads']), 'aborted' in d, {k:(round(v['ms_per_step'],4), v['bytes_per_gpu_per_step']) for k,v in gm.items()})" || { tail -5 gpurun_out/one_rank_$via.err; exit 1; }
done
