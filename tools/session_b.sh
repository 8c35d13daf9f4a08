#!/bin/bash
# Round-3 session B (GPU box): the whole GPU suite on the new build, then the default bench line.
out=gpurun_out/r3b; mkdir -p $out; export TMPDIR=/tmp
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $out/suite.txt 2>&1; rc=$?
tail -15 $out/suite.txt
[ $rc -ne 0 ] && { echo "SUITE FAILED rc=$rc"; grep -n "Error\|error\|assert" $out/suite.txt | head -40; exit 1; }
python bench.py > $out/bench_default.json 2> $out/bench_default.err; echo "bench rc=$?"; tail -3 $out/bench_default.err
python -c "
import json; d=json.loads(open('$out/bench_default.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms/step', d['ms_per_step'], d['kernels_ms'], d['roofline']['kernel'], d['roofline']['frac'])
print('cpu', {k: v for k, v in d['cpu_baseline'].items() if k not in ('sample',)})
for c in d['configs']: print(c['workload'][:12], c['ms_per_step'], c['kernels_ms'])
"
