#!/usr/bin/env python3
"""One iteration of the 20-iteration moment segmentation (or one moment-retrieval batch) as the GPU saw it: the kernels between two
launches of joint_mask_add_kernel (once per encoder pass) near the end of a rocprofv3 --kernel-trace CSV.
   python tools/iter_timeline.py x_kernel_trace.csv"""
import csv, re, sys
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))))
marks = [i for i, r in enumerate(rows) if "joint_mask_add_kernel" in r[2]]
a, b = marks[-4], marks[-3]
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"^void ", "", n); n = re.sub(r"at::native::", "", n)
    return re.sub(r"\(.*", "", n)[:60]
busy = 0
for i in range(a, b):
    s, e, n = rows[i]
    busy += e - s
    print(f"{short(n):62s} {(e - s) / 1e3:7.1f} us   gap before {(s - rows[i - 1][1]) / 1e3:6.1f} us")
print(f"# {b - a} kernels, {(rows[b][0] - rows[a][0]) / 1e3:.0f} us start to start, busy {busy / 1e3:.0f} us")
