"""Per-kernel SQ / GRBM counter summary of one rocprofv3 --pmc pass (averages per launch).
MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (kernel cycles x 1024 SIMDs); GRBM_GUI_ACTIVE is summed over the
8 XCDs, so kernel cycles = GRBM_GUI_ACTIVE / 8.  The three wave-cycle buckets are disjoint (MI355X_MICROARCH.md).
usage: python tools/sq_summary.py <pmc dir> <out.md> "<command line>" """
import collections, csv, glob, re, sys

d, out_md, cmd = sys.argv[1:4]
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if "stdadk::" not in n:
        continue
    short = re.sub(r"<.*", "", re.sub(r"^void ", "", n).split("(")[0].replace("stdadk::", ""))
    a = acc[short][r["Counter_Name"]]
    a[0] += 1
    a[1] += float(r["Counter_Value"])
lines = ["# SQ / GRBM counters per launch (rocprofv3 PMC, one pass)", "", f"Command: `{cmd}`", "",
         "| kernel | launches | kernel cycles (GUI_ACTIVE/8) | MFMA busy cycles | MFMA util | waves: issuing | parked "
         "(s_waitcnt / barrier) | issue-stalled |", "|---|---|---|---|---|---|---|---|"]
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]["GRBM_GUI_ACTIVE"][1]):
    g = {c: t / n for c, (n, t) in v.items()}
    cyc = g["GRBM_GUI_ACTIVE"] / 8
    wc = max(g["SQ_WAVE_CYCLES"], 1.0)
    lines.append(f"| `{k}` | {v['GRBM_GUI_ACTIVE'][0]} | {cyc:.0f} | {g['SQ_VALU_MFMA_BUSY_CYCLES']:.0f} | "
                 f"{g['SQ_VALU_MFMA_BUSY_CYCLES'] / (cyc * 1024) * 100:.1f} % | {g['SQ_ACTIVE_INST_ANY'] / wc * 100:.0f} % | "
                 f"{g['SQ_WAIT_ANY'] / wc * 100:.0f} % | {g['SQ_WAIT_INST_ANY'] / wc * 100:.0f} % |")
open(out_md, "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
