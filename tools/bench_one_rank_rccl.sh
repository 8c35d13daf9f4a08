#!/bin/bash
# The N > 1 code path of bench.py with ONE rank and the real RCCL backend (run on the GPU box): the in-place all-gather of
# every payload through torch.distributed, through the C-ABI's rc_gather_trajectory, and the peer-copy transport (with one
# rank: the local copy and the flag protocol without peers), with the driver's own step counts.
mkdir -p gpurun_out
for via in torch abi p2p; do
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py \
      --gpus 1 --force-gather --steps 20 --warmup 5 --gather-via $via 2>gpurun_out/one_rank_$via.err | \
    python -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('$via', round(d['ms_per_step'],4), d['config']['gather'], 'rccl_ranks', d['config']['rccl_ranks'], 'abi', d['config']['abi_comm_ranks'], 'gather_check', d['gather_check']['ok'], sorted(d['gather_check']['payloads']), {k:(round(v['ms_per_step'],4), v['bytes_per_gpu_per_step']) for k,v in d['gather_modes'].items()})" || { tail -5 gpurun_out/one_rank_$via.err; exit 1; }
done
