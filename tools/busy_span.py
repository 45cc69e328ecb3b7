#!/usr/bin/env python3
"""GPU-busy time against wall time of a rocprofv3 --kernel-trace CSV: sum of kernel durations, first start to last end of the last
`frac` of the trace (steady state), kernels and idle gaps — is a loop GPU- or host-bound?   python tools/busy_span.py x_kernel_trace.csv [0.5]"""
import csv, sys
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))))
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
rows = rows[int(len(rows) * (1 - frac)):]
busy = sum(e - s for s, e, _ in rows)
span = rows[-1][1] - rows[0][0]
gaps = [rows[i][0] - rows[i - 1][1] for i in range(1, len(rows))]
big = sum(g for g in gaps if g > 10000)
print(f"{len(rows)} kernels over {span / 1e6:.2f} ms: busy {busy / 1e6:.2f} ms ({100.0 * busy / span:.0f} %), gaps > 10 us {big / 1e6:.2f} ms, "
      f"mean kernel {busy / len(rows) / 1e3:.1f} us")
