#!/bin/bash
# A/B on one box: shipped scan vs the direction-table timing experiment (RC_EXP_DIR_TABLE)
out=gpurun_out/r3d; mkdir -p $out
lib=racing_dreamer_amd/lib/libracecar_hip.so; cp $lib /tmp/orig.so
echo "--- scan A/B: shipped vs direction table (timing experiment)" | tee $out/ab_dirtable.txt
for r in 1 2 3; do for v in shipped dirtable; do
  cp racing_dreamer_amd/lib/ab/$v.so $lib
  python bench.py --no-cpu-baseline --no-ftg --no-configs --steps 300 --warmup 30 2>/dev/null | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('rep $r  %-10s scan %.4f ms  step %.4f ms' % ('$v', d['roofline']['avg_launch_ms'], d['ms_per_step']))" | tee -a $out/ab_dirtable.txt
done; done
echo "--- 4096 cars (columbia)" | tee -a $out/ab_dirtable.txt
for r in 1 2; do for v in shipped dirtable; do
  cp racing_dreamer_amd/lib/ab/$v.so $lib
  python bench.py --no-cpu-baseline --no-ftg --no-configs --envs 4096 --track columbia --steps 1000 --warmup 100 2>/dev/null | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('rep $r  %-10s scan %.4f ms  step %.4f ms' % ('$v', d['roofline']['avg_launch_ms'], d['ms_per_step']))" | tee -a $out/ab_dirtable.txt
done; done
cp /tmp/orig.so $lib
