import csv, sys, re
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))))
idx = [i for i, r in enumerate(rows) if "beam_backtrack" in r[2]]
a = idx[-2]
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"^void ", "", n); n = re.sub(r"at::native::", "", n)
    return re.sub(r"\(.*", "", n)[:56]
t0 = rows[a][1]
for i in range(a, a + 70):
    s, e, n = rows[i]
    print(f"{short(n):58s} {(e - s) / 1e3:7.1f} us   gap before {(s - rows[i - 1][1]) / 1e3:7.1f} us   t = {(s - t0) / 1e3:8.1f}")
