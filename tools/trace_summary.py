#!/usr/bin/env python3
"""Per-problem summary of a rocprofv3 --kernel-trace CSV: one kernel name (e.g. gemm_p256<1, 64, false>) serves the vision
tower's and the text tower's shapes, which rocprofv3 --stats averages together; this splits each name's dispatches into
duration clusters (a new cluster where the sorted durations jump by > 40 %) so every line is one problem shape.
   python tools/trace_summary.py gpurun_out/prof/r_kernel_trace.csv > profiles/rNN/rocprofv3_kernel_trace_by_shape.csv"""
import collections
import csv
import sys

rows = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    rows[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["VGPR_Count"], r["Accum_VGPR_Count"],
                                   r["LDS_Block_Size"], r["Scratch_Size"], r["Grid_Size_X"], r["Workgroup_Size_X"]))
out = []
for name, ds in rows.items():
    ds.sort()
    cluster = [ds[0]]
    for d in ds[1:] + [None]:
        if d is not None and d[0] <= cluster[-1][0] * 1.4:
            cluster.append(d)
            continue
        t = [c[0] for c in cluster]
        out.append((sum(t), name, len(t), sum(t) / len(t), t[0], t[-1]) + cluster[0][1:])
        cluster = [d] if d is not None else []
out.sort(reverse=True)
total = sum(o[0] for o in out)
w = csv.writer(sys.stdout)
w.writerow(["Name", "Calls", "AverageNs", "MinNs", "MaxNs", "Percentage", "VGPR", "AccumVGPR", "LDS_Bytes", "Scratch", "Grid_X", "Workgroup_X"])
for tot, name, n, avg, lo, hi, vg, av, lds, scr, gx, wx in out:
    if tot / total < 2e-4:
        continue
    w.writerow([name[:110], n, f"{avg:.0f}", lo, hi, f"{100 * tot / total:.2f}", vg, av, lds, scr, gx, wx])
