"""Summaries of rocprofv3 rocpd (.db) output, in the layout of its CSV reports.

    python tools/rocpd_summary.py stats <results.db>      # kernel_stats.csv columns
    python tools/rocpd_summary.py pmc <results.db> ...    # kernel, counter, mean value per dispatch (summed over instances)
"""
import sqlite3
import sys
from collections import defaultdict


def stats(db):
    c = sqlite3.connect(db)
    rows = defaultdict(list)
    for name, dur in c.execute("select name, duration from kernels"):
        rows[name].append(dur)
    total = sum(sum(v) for v in rows.values())
    out = ['"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs","StdDev"']
    for name, v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
        n, s = len(v), sum(v)
        mean = s / n
        sd = (sum((x - mean) ** 2 for x in v) / (n - 1)) ** 0.5 if n > 1 else 0.0
        out.append(f'"{name}",{n},{s},{mean:.6f},{100.0 * s / total:.2f},{min(v)},{max(v)},{sd:.6f}')
    return "\n".join(out)


def pmc(dbs):
    acc = defaultdict(lambda: defaultdict(float))
    for db in dbs:
        c = sqlite3.connect(db)
        for kernel, disp, counter, value in c.execute(
                "select kernel_name, dispatch_id, counter_name, value from counters_collection"):
            acc[(kernel, counter)][(db, disp)] += value          # one row per hardware instance: sum them
    out = ["kernel,counter,mean_per_dispatch,dispatches"]
    for (kernel, counter), d in sorted(acc.items()):
        out.append(f'"{kernel}",{counter},{sum(d.values()) / len(d):.1f},{len(d)}')
    return "\n".join(out)


def _short(name):
    """`rc_raycast_car_kernel<1>` from `void (anonymous namespace)::rc_raycast_car_kernel<1>(RcParams, int)`."""
    name = name.replace("void ", "").replace("(anonymous namespace)::", "")
    return name.split("(")[0][-48:]


def gaps(db):
    """Idle time on the device between consecutive kernels (end of one to start of the next), by successor."""
    c = sqlite3.connect(db)
    rows = list(c.execute("select name, start, end from kernels order by start"))
    acc = defaultdict(list)
    for (n0, s0, e0), (n1, s1, e1) in zip(rows, rows[1:]):
        acc[(_short(n0), _short(n1))].append(s1 - e0)
    out = ["prev,next,count,mean_gap_ns,median_gap_ns"]
    for k, v in sorted(acc.items(), key=lambda kv: -len(kv[1])):
        v = sorted(v)
        out.append(f'"{k[0]}","{k[1]}",{len(v)},{sum(v) / len(v):.0f},{v[len(v) // 2]}')
    return "\n".join(out)


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        print(stats(sys.argv[2]))
    elif sys.argv[1] == "gaps":
        print(gaps(sys.argv[2]))
    else:
        print(pmc(sys.argv[2:]))
