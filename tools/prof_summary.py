"""Print a per-kernel summary (per-step microseconds) of a rocprofv3 --kernel-trace --stats run.
usage: python tools/prof_summary.py <dir containing *_kernel_stats.csv> [steps] [--md title]"""
import csv, glob, sys
d = sys.argv[1]
steps = float(sys.argv[2]) if len(sys.argv) > 2 and not sys.argv[2].startswith("--") else None
f = glob.glob(d + "/**/*_kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot/1e6:.3f} ms over {len(rows)} kernels" + (f"; {tot/1e3/steps:.1f} us per step over {steps:.0f} steps" if steps else ""))
print("| kernel | calls | avg us | total us | % |")
print("|---|---|---|---|---|")
for r in rows[:40]:
    print(f"| `{r['Name'][:90]}` | {r['Calls']} | {float(r['AverageNs'])/1e3:.1f} | {float(r['TotalDurationNs'])/1e3:.1f} | {float(r['Percentage']):.2f} |")
