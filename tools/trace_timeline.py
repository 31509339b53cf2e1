"""Per-step timeline from a rocprofv3 --kernel-trace csv: kernel durations and the gaps between
consecutive kernels of the main stream (largest-queue heuristic), for the last N steps.
usage: python tools/trace_timeline.py <dir> [anchor kernel substring, default adamw_ema]"""
import csv, glob, sys
d = sys.argv[1]
anchor = sys.argv[2] if len(sys.argv) > 2 else "adamw_ema"
f = glob.glob(d + "/**/*_kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
q = {}
for r in rows:
    q[r["Queue_Id"]] = q.get(r["Queue_Id"], 0) + 1
mainq = max(q, key=q.get)
print("queues:", q, "main:", mainq)
main = [r for r in rows if r["Queue_Id"] == mainq]
ends = [i for i, r in enumerate(main) if anchor in r["Kernel_Name"]]
ends = ends[-40:-34]
for a, b in zip(ends[:-1], ends[1:]):
    seg = main[a + 1:b + 1]
    t0 = int(main[a]["End_Timestamp"])
    wall = int(seg[-1]["End_Timestamp"]) - t0
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg)
    line = []
    prev = t0
    for r in seg:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        nm = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("stdadk::", "")[:18]
        line.append(f"{nm}:{(e - s) / 1e3:.1f}(+{(s - prev) / 1e3:.1f})")
        prev = e
    print(f"step wall {wall / 1e3:.1f} us, kernels {busy / 1e3:.1f} us | " + " ".join(line))
side = [r for r in rows if r["Queue_Id"] != mainq and "stdadk" in r["Kernel_Name"]]
if side:
    agg = {}
    for r in side[-200:]:
        nm = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("stdadk::", "")[:30]
        a = agg.setdefault(nm, [0, 0])
        a[0] += 1; a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    print("side stream:", {k: round(v[1] / v[0] / 1e3, 1) for k, v in agg.items()})
