#!/usr/bin/env python3
"""One decoded word of step captioning as the GPU saw it: the kernels between two launches of the decode step's first kernel
(fill_i32_kernel) near the middle of a rocprofv3 --kernel-trace CSV, with each one's duration and the idle gap before it.
   python tools/word_timeline.py gpurun_out/capt/x_kernel_trace.csv [which_word]"""
import csv, re, sys
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))))
marks = [i + 1 for i, r in enumerate(rows) if "tail_select_kernel" in r[2] or "beam_advance_kernel" in r[2]]   # a word ends with the beam bookkeeping
w = int(sys.argv[2]) if len(sys.argv) > 2 else len(marks) // 2
a, b = marks[w], marks[w + 1]
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"^void ", "", n)
    return re.sub(r"\(.*", "", n)[:44]
busy = 0
print(f"# word {w}: {b - a} kernels, {(rows[b][0] - rows[a][0]) / 1e3:.1f} us from first start to the next word's first start")
for i in range(a, b):
    s, e, n = rows[i]
    gap = s - rows[i - 1][1] if i > 0 else 0
    busy += e - s
    print(f"{short(n):46s} {(e - s) / 1e3:7.1f} us   gap before {gap / 1e3:6.1f} us")
print(f"# kernels busy {busy / 1e3:.1f} us")
