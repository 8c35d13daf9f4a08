#!/bin/bash
# Round-4 closing session (GPU box): the whole GPU suite, the default bench line and the line at the driver's flags, the N > 1
# path with one RCCL rank through all three transports, where its timed window goes, smoke().
out=gpurun_out/r4f; mkdir -p $out; export TMPDIR=/tmp
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $out/suite.txt 2>&1; rc=$?
tail -6 $out/suite.txt
[ $rc -ne 0 ] && { echo "SUITE FAILED rc=$rc"; grep -n "Error\|error\|assert" $out/suite.txt | head -40; exit 1; }
python bench.py > $out/bench_default.json 2> $out/bench_default.err; echo "bench rc=$?"
python bench.py --steps 20 --warmup 5 > $out/bench_driver_flags.json 2>> $out/bench_default.err; echo "bench (driver's flags) rc=$?"
python -c "
import json
for f in ('bench_default', 'bench_driver_flags'):
    d=json.loads(open('$out/' + f + '.json').read().strip().splitlines()[-1])
    print(f, 'value', round(d['value']/1e6, 1), 'M  ms/step', round(d['ms_per_step'], 4), d['kernels_ms'], d['roofline']['kernel'], round(d['roofline']['frac'], 4), 'issue', d['roofline']['issue_frac'], 'tracks', [(t['track'], t['raycast_ms']) for t in d['tracks']], 'fresh', d['fresh_reset']['raycast_ms'], 'cpu x', round(d['cpu_baseline']['speedup_all_over_one_thread'], 2), 'on', d['cpu_baseline']['cores_effective'], 'errors', d.get('leg_errors'))
"
bash tools/bench_one_rank_rccl.sh | tee $out/bench_one_rank_rccl.txt
RC_BENCH_TRACE_TIMED=1 bash tools/fixed_cost.sh | tee $out/fixed_cost.txt
python -c 'import __graft_entry__ as g; g.smoke()' 2>&1 | tail -2
