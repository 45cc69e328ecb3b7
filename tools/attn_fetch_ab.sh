R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p $R/gpurun_out/attn
cd /tmp
for m in 0 1; do
  timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/attn/fetch_m$m -o p -- python $R/tools/attn_bench.py --variants 5 --skew 12 --iters 3 --map $m > $R/gpurun_out/attn/fetch_m$m.log 2>&1
done
cd $R
python - <<'PY'
import csv, glob
for m in (0, 1):
    v = []
    for f in glob.glob(f"gpurun_out/attn/fetch_m{m}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == "FETCH_SIZE" and "attention_kernel_v3" in r["Kernel_Name"]:
                v.append(float(r["Counter_Value"]))
    print(f"map {m}: {len(v)} launches, fetch {2 * 1024 * sum(v) / max(len(v), 1) / 1e9:.2f} GB per launch (FETCH_SIZE x2, KiB)")
PY
