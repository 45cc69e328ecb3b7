#!/usr/bin/env python3
"""One joint-model training step as the GPU saw it: the kernels between two launches of the step's first library kernel
(joint_time_kernel runs once per forward) near the end of a rocprofv3 --kernel-trace CSV, each with its duration and the idle gap
before it, then totals by kernel.     python tools/step_timeline.py x_kernel_trace.csv [--list]"""
import csv, re, sys
from collections import defaultdict
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))))
marks = [i for i, r in enumerate(rows) if "joint_time_kernel" in r[2]]
a, b = marks[-3], marks[-2]
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"^void ", "", n); n = re.sub(r"at::native::", "", n)
    return re.sub(r"\(.*", "", n)[:60]
busy, gaps, big = 0, 0, 0
by = defaultdict(lambda: [0, 0])
for i in range(a, b):
    s, e, n = rows[i]
    gap = s - rows[i - 1][1]
    busy += e - s; gaps += max(gap, 0); big += gap if gap > 10000 else 0
    by[short(n)][0] += 1; by[short(n)][1] += e - s
    if "--list" in sys.argv:
        print(f"{short(n):62s} {(e - s) / 1e3:7.1f} us   gap before {gap / 1e3:6.1f} us")
print(f"# step: {b - a} kernels, {(rows[b][0] - rows[a][0]) / 1e3:.0f} us start to start; busy {busy / 1e3:.0f} us, idle {gaps / 1e3:.0f} us "
      f"(of which gaps > 10 us: {big / 1e3:.0f} us)")
for n, (c, t) in sorted(by.items(), key=lambda kv: -kv[1][1]):
    print(f"{c:4d} x {t / c / 1e3:7.1f} us = {t / 1e3:7.0f} us  {n}")
