#!/bin/bash
# Kernel timeline of the first 20 steps after a reset against 40 steady ones (GPU box): bash tools/fresh_window.sh
R=$(pwd); export TMPDIR=/tmp; mkdir -p gpurun_out/sh; rm -rf gpurun_out/sh/fw
(cd /tmp && rocprofv3 --kernel-trace -d $R/gpurun_out/sh/fw -o fw -- python3 $R/tools/fresh_window_trace.py $1) 2>/dev/null | grep "fresh window"
python - <<PY
import sqlite3, glob, re
db=glob.glob("gpurun_out/sh/fw/**/*.db", recursive=True)[0]
c=sqlite3.connect(db)
rows=[(re.sub(r"void \(anonymous namespace\)::|\(.*", "", nm), s, e) for nm,s,e in c.execute("select name, start, end from kernels order by start")]
# the window of interest: after the LAST reset kernel
last_reset=max(i for i,(n,s,e) in enumerate(rows) if "rc_reset_kernel" in n)
win=rows[last_reset:]
t0=win[0][1]
print("kernels from the last reset on (start ms, duration us, gap to the previous kernel's end us):")
prev=None
for i,(n,s,e) in enumerate(win[:12]):
    print("%3d %-48s start %8.3f  dur %7.1f  gap %6.1f" % (i, n[:48], (s-t0)/1e6, (e-s)/1e3, 0.0 if prev is None else (s-prev)/1e3))
    prev=e
def summarize(seg, label):
    scans=[(s,e) for n,s,e in seg if "raycast_car" in n]
    dyn=[(s,e) for n,s,e in seg if "dynamics" in n]
    other=[(n,s,e) for n,s,e in seg if "raycast_car" not in n and "dynamics" not in n]
    span=(seg[-1][2]-seg[0][1])/1e6
    busy=sum(e-s for n,s,e in seg)/1e6
    print("%s: %d kernels over %.3f ms, busy %.3f ms, scans %d avg %.1f us, dynamics %d avg %.1f us, other kernels %d (%.1f us in all): %s" % (
        label, len(seg), span, busy, len(scans), sum(e-s for s,e in scans)/len(scans)/1e3, len(dyn), sum(e-s for s,e in dyn)/max(len(dyn),1)/1e3,
        len(other), sum(e-s for n,s,e in other)/1e3, sorted(set(n for n,_,_ in other))))
# split: first 20 steps after the reset's own scan; then the 40 steady steps
steps=[i for i,(n,s,e) in enumerate(win) if "dynamics" in n]
first=win[steps[0]:steps[20]]
steady=win[steps[20]:steps[20]+ (steps[59]-steps[20]) + 2]
summarize(first, "first 20 steps")
summarize(steady, "next 40 steps")
PY
rm -rf gpurun_out/sh/fw
