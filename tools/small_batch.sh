#!/bin/bash
# Do the stamps of consecutive kernels overlap at a small batch?  (GPU box): bash tools/small_batch.sh
R=$(pwd); export TMPDIR=/tmp; mkdir -p gpurun_out/sh; rm -rf gpurun_out/sh/sb
(cd /tmp && rocprofv3 --kernel-trace -d $R/gpurun_out/sh/sb -o sb -- python3 $R/tools/small_batch_trace.py) > /dev/null 2>&1
python - <<PY
import sqlite3, glob, re
db=glob.glob("gpurun_out/sh/sb/**/*.db", recursive=True)[0]
c=sqlite3.connect(db)
rows=[(re.sub(r"void \(anonymous namespace\)::|\(.*", "", nm), s, e) for nm,s,e in c.execute("select name, start, end from kernels order by start")]
steps=[i for i,(n,s,e) in enumerate(rows) if "dynamics" in n]
seg=rows[steps[-201]:steps[-1]]          # the last 200 whole steps
span=(seg[-1][2]-seg[0][1])/1e3
dur=sum(e-s for n,s,e in seg)/1e3
gaps=[(seg[i+1][1]-seg[i][2])/1e3 for i in range(len(seg)-1)]
scan=[e-s for n,s,e in seg if "raycast" in n]; dyn=[e-s for n,s,e in seg if "dynamics" in n]
other=sorted(set(n for n,s,e in seg if "raycast" not in n and "dynamics" not in n))
print("200 steps at 4 096 envs (columbia): span %.1f us = %.2f us per step; sum of the kernels' own stamps %.1f us = %.2f per step (scan %.2f, dynamics %.2f, others %s)" % (
    span, span/200, dur, dur/200, sum(scan)/len(scan)/1e3, sum(dyn)/len(dyn)/1e3, other))
print("gap between one kernel's end stamp and the next one's start stamp: mean %.2f us, min %.2f, max %.2f; negative (overlapping stamps) in %d of %d" % (
    sum(gaps)/len(gaps), min(gaps), max(gaps), sum(g < 0 for g in gaps), len(gaps)))
PY
rm -rf gpurun_out/sh/sb
