#!/usr/bin/env python3
"""Concurrency picture of a rocprofv3 --kernel-trace of bench.py with frames in flight:
for the timed region (last 60 % of the trace) prints, per kernel, the mean duration under overlap, and the fraction of
wall time during which 0 / 1 / 2 / 3+ kernels were running and during which at least one k_render_bwd was running.
usage: trace_overlap.py DIR_WITH_kernel_trace.csv"""
import csv
import glob
import sys
from collections import defaultdict


def main():
    path = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = [r for r in csv.DictReader(open(path)) if r["Kernel_Name"].startswith(("k_", "void k_"))]
    ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0]) for r in rows]
    ev.sort()
    t_lo = ev[int(len(ev) * 0.4)][0]
    ev = [e for e in ev if e[0] >= t_lo]
    t0, t1 = ev[0][0], max(e[1] for e in ev)
    dur = defaultdict(list)
    for s, e, n in ev:
        dur[n].append(e - s)
    print("window %.2f ms, %d kernels" % ((t1 - t0) / 1e6, len(ev)))
    for n, d in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
        print("  %-28s n=%4d mean %7.1f us  sum/window %.2f" % (n[:28], len(d), sum(d) / len(d) / 1e3, sum(d) / (t1 - t0)))
    pts = []
    for s, e, n in ev:
        pts.append((s, 1, n))
        pts.append((e, -1, n))
    pts.sort()
    conc, bwd, last = 0, 0, t0
    hist, bwd_time = defaultdict(int), 0
    for t, d, n in pts:
        hist[min(conc, 4)] += t - last
        if bwd > 0:
            bwd_time += t - last
        last = t
        conc += d
        if "render_bwd" in n:
            bwd += d
    tot = t1 - t0
    print("  kernels running at once: " + "  ".join("%d: %.2f" % (k, v / tot) for k, v in sorted(hist.items())))
    print("  >=1 k_render_bwd running: %.2f of the window" % (bwd_time / tot))


if __name__ == "__main__":
    main()
