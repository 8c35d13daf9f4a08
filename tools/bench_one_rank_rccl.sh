#!/bin/bash
# The plain `--gpus N` line's new legs (whole-record gathers at action_repeat 4, configs[4]'s track mix) with ONE rank over
# the real RCCL backend (GPU box): bash tools/bench_one_rank_rccl.sh
mkdir -p gpurun_out
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29513 bench.py \
    --gpus 1 --force-gather --steps 20 --warmup 5 2>gpurun_out/one_rank_r5.err > gpurun_out/one_rank_r5.json || { tail -5 gpurun_out/one_rank_r5.err; exit 1; }
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/one_rank_r5.json").read().splitlines() if l.startswith("{")][-1])
gm = d["gather_modes"]
print("headline", d["config"]["gather"], round(d["ms_per_step"], 4), "ms/step; steady", round(gm["sharded"]["steady_state"]["ms_per_step"], 4),
      "rccl_ranks", d["config"]["rccl_ranks"], "aborted" in d, d.get("leg_errors"))
print("per sub-step:", {k: (round(v["ms_per_step"], 4), v["bytes_per_gpu_per_step"]) for k, v in gm.items()})
print("per agent step (repeat 4):", {k: (round(v["ms_per_agent_step"], 4), v["bytes_per_gpu_per_agent_step"], round(v["link_bound_ms_per_agent_step"], 3),
                                        round(v["agent_steps_per_s"] / 1e6, 1), v["check"]["ok"]) for k, v in d["gather_modes_repeat_4"].items()})
print("configs4_track_mix:", {k: v for k, v in d["configs4_track_mix"].items() if k != "workload"})
print("gather_check", d["gather_check"]["ok"], sorted(d["gather_check"]["payloads"]))
PY
# ... and the per-step record gathers through the two other transports (the headline stays the sharded store over torch.distributed)
for via in abi p2p; do
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29514 bench.py \
      --gpus 1 --force-gather --steps 20 --warmup 5 --gather-via $via 2>gpurun_out/one_rank_$via.err > gpurun_out/one_rank_$via.json || { tail -5 gpurun_out/one_rank_$via.err; exit 1; }
  python - <<PY
import json
d = json.loads([l for l in open("gpurun_out/one_rank_$via.json").read().splitlines() if l.startswith("{")][-1])
gm = d["gather_modes"]
print("via $via:", {k: (round(v["ms_per_step"], 4), v.get("check", {}).get("via"), v.get("check", {}).get("ok")) for k, v in gm.items()}, "abi ranks", d["config"]["abi_comm_ranks"],
      "aborted" in d, d.get("leg_errors"), d.get("legs_skipped"))
PY
done
