#!/usr/bin/env python3
"""Markdown summary of tools/pmc_sq.sh output files (per-kernel averages of SQ / TCC counters collected in separate
rocprofv3 --pmc passes): matrix-pipe duty cycle, wait shares, LDS activity, L2 hit rate.
   python tools/pmc_summary.py profiles/r02/pmc_bench_counters.txt [more files] > profiles/r02/pmc_summary.md
Derivations: SQ_VALU_MFMA_BUSY_CYCLES counts SIMD-cycles (= 16 x SQ_INSTS_MFMA for 16x16x32 bf16); GRBM_GUI_ACTIVE is summed
over the 8 XCDs, so cycles per XCD = GRBM / 8 and the chip has 1024 SIMDs: duty = MFMA_BUSY / (GRBM / 8 * 1024).  The clock
estimate divides the GRBM pass's cycles by the trace pass's duration (different passes: +-10 %)."""
import re
import sys

rows = []
for path in sys.argv[1:]:
    txt = open(path).read()
    for blk in re.split(r"\n(?=\S)", txt):
        m = re.match(r"(.*?)\| dispatches (\d+) avg ([\d.]+) us", blk)
        if not m:
            continue
        c = {k: float(v) for k, v in re.findall(r"\s+(\w+)\s+([\d.e+]+)\n", blk + "\n")}
        if "GRBM_GUI_ACTIVE" not in c or "SQ_VALU_MFMA_BUSY_CYCLES" not in c:
            continue
        name = re.sub(r"\(anonymous namespace\)::|void |\(.*", "", m.group(1)).strip()[:70]
        cyc, dur, wc = c["GRBM_GUI_ACTIVE"] / 8, float(m.group(3)), c["SQ_WAVE_CYCLES"]
        rows.append((name, int(m.group(2)), dur, cyc / dur / 1e3, 100 * c["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024),
                     100 * c["SQ_WAIT_ANY"] / wc, 100 * c["SQ_WAIT_INST_ANY"] / wc, 100 * c["SQ_ACTIVE_INST_ANY"] / wc,
                     100 * c["SQ_LDS_IDX_ACTIVE"] / (cyc * 256), 100 * c["SQ_LDS_BANK_CONFLICT"] / max(c["SQ_LDS_IDX_ACTIVE"], 1),
                     100 * c["TCC_HIT_sum"] / max(c["TCC_HIT_sum"] + c["TCC_MISS_sum"], 1), c["TCC_EA0_RDREQ_sum"] * 64 * 2 / 1e9))
print("| kernel | launches | avg us (traced pass) | ~GHz | MFMA duty % | wave cycles in s_waitcnt/barrier % | issue-stalled % | issuing % | "
      "LDS array active % | LDS bank-conflict % of LDS cycles | L2 hit % | memory-side read GB / launch (RDREQ x 128 B) |")
print("|---|---|---|---|---|---|---|---|---|---|---|---|")
for r in rows:
    print("| `%s` | %d | %.1f | %.2f | %.1f | %.1f | %.1f | %.1f | %.1f | %.1f | %.0f | %.2f |" % r)
