#!/bin/bash
# One rank over real RCCL on one box, alternating: the headline's env created AFTER the communicator (what bench.py does) against
# the env of stage 1 kept (RC_EXP_KEEP_ENV=1: its stream is older than the communicator's).  profiles/r06_i_ab_stream_order_one_rank.txt
run() {  # name, env...
  name=$1; shift
  env "$@" python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py \
      --gpus 1 --force-gather --steps 20 --warmup 5 --no-gather-modes 2>gpurun_out/abso_$name.err > gpurun_out/abso_$name.json || { tail -3 gpurun_out/abso_$name.err; }
  python - <<PY
import json
d = json.loads([l for l in open("gpurun_out/abso_$name.json").read().splitlines() if l.startswith("{")][-1])
print("$name headline", round(d["ms_per_step"], 4), "steady", round(d["gather_modes"]["sharded"].get("steady_state", {}).get("ms_per_step", 0), 4), "scan", d["kernels_ms"].get("rc_raycast_kernel"))
PY
}
for rep in 1 2 3; do
  run env_after_communicator A=1
  run env_before_communicator RC_EXP_KEEP_ENV=1
done
