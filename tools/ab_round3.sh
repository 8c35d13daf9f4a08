#!/bin/bash
# Round-3 session A (run on the GPU box; the record is profiles/r03_a_*; needs the builds of that session in
# racing_dreamer_amd/lib/ab/ and the round-2 render kernel, patch_variant 8, which has been removed since): parity of the new default build, the trip-budget termination check, then A/B on
# this one box of (a) the scan with and without the in-shadow trip budget, (b) the lidar_occupancy render: round-2 kernel
# (patch_variant 8), one wave per car with non-temporal (0) and plain (2) stores; then the render's HBM write traffic.
out=gpurun_out/r3a; mkdir -p $out; export TMPDIR=/tmp
lib=racing_dreamer_amd/lib/libracecar_hip.so
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py tests/test_gpu_gather.py -m gpu -x -q > $out/parity.txt 2>&1; rc=$?
tail -5 $out/parity.txt
[ $rc -ne 0 ] && { echo "PARITY FAILED rc=$rc"; exit 1; }
echo "--- trip budget: band 2^-30 on the production scan must terminate" | tee $out/budget.txt
timeout -k 10 120 python - >> $out/budget.txt 2>&1 <<'PY' || { echo "BUDGET CHECK FAILED"; tail -5 $out/budget.txt; exit 1; }
import time, torch
from racing_dreamer_amd.batched_env import BatchedRaceEnv
env = BatchedRaceEnv("austria", 65536, 1, auto_reset=True)
env.reset(mode="random", seed=0)
for k in range(30):
    env.step_random(seed=1, step=k)
env.sync()
ref = env.views["lidar"].clone()
env.debug_set("band_log2", -30)
t0 = time.perf_counter()
env.step_random(seed=1, step=30)
env.sync()
dt = time.perf_counter() - t0
l = env.views["lidar"]
print(f"band 2^-30: one step of 65 536 cars took {dt*1e3:.1f} ms; rays reading 15.0 (no return): {(l == 15.0).float().mean().item():.4f} "
      f"(with the shipped band: {(ref == 15.0).float().mean().item():.4f})")
env.debug_set("band_log2", 0)
env.step_random(seed=1, step=31)
env.sync()
print("back on the shipped band: finite", bool(torch.isfinite(env.views['lidar']).all().item()))
PY
tail -3 $out/budget.txt
echo "--- scan A/B" | tee $out/ab_scan.txt
for r in 1 2 3; do for v in bound0 bound2; do
  cp racing_dreamer_amd/lib/ab/$v.so $lib
  python bench.py --no-cpu-baseline --no-ftg --no-configs --steps 300 --warmup 30 2>/dev/null | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('rep $r  %-10s scan %.4f ms  step %.4f ms' % ('$v', d['roofline']['avg_launch_ms'], d['ms_per_step']))" | tee -a $out/ab_scan.txt
done; done
echo "--- scan A/B at 4096 cars (columbia)" | tee -a $out/ab_scan.txt
for r in 1 2; do for v in bound0 bound2; do
  cp racing_dreamer_amd/lib/ab/$v.so $lib
  python bench.py --no-cpu-baseline --no-ftg --no-configs --envs 4096 --track columbia --steps 1000 --warmup 100 2>/dev/null | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('rep $r  %-10s scan %.4f ms  step %.4f ms' % ('$v', d['roofline']['avg_launch_ms'], d['ms_per_step']))" | tee -a $out/ab_scan.txt
done; done
cp racing_dreamer_amd/lib/ab/bound2.so $lib
echo "--- patch A/B (8 = round-2 kernel, 0 = wave per car NT stores, 2 = wave per car plain stores)" | tee $out/ab_patch.txt
for r in 1 2 3; do for v in 8 0 2; do
  python bench.py --no-cpu-baseline --no-ftg --no-configs --obs-type lidar_occupancy --steps 100 --warmup 10 --debug-knob patch_variant=$v 2>/dev/null | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('rep $r  patch_variant $v  step %.4f ms ' % d['ms_per_step'], d['kernels_ms'])" | tee -a $out/ab_patch.txt
done; done
echo "--- patch HBM traffic" | tee $out/pmc_patch.txt
R=$(pwd)
for v in 8 0 2; do
  for group in "WRITE_SIZE" "FETCH_SIZE"; do
    rm -rf $out/pass; (cd /tmp && rocprofv3 --pmc $group -d $R/$out/pass -o pmc -- python3 $R/bench.py --no-cpu-baseline --no-ftg --no-configs --obs-type lidar_occupancy --steps 10 --warmup 2 --debug-knob patch_variant=$v > /dev/null 2> $R/$out/err_pmc.log) || { echo "rocprof failed"; tail -3 $out/err_pmc.log; exit 1; }
    echo "patch_variant $v $group" | tee -a $out/pmc_patch.txt
    python tools/rocpd_summary.py pmc $(find $out/pass -name "*.db" | sort) | grep -i "patch" | tee -a $out/pmc_patch.txt
  done
done
rm -rf $out/pass
