"""Wall time of every step (AdamW end -> next AdamW end) from a rocprofv3 --kernel-trace csv of a short bench run:
shows where a 20-step timed region loses time against a 200-step one (first steps after the synchronise, or all).
usage: python tools/trace_steps.py <dir> [first n steps, default 40]"""
import csv, glob, sys
d = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
f = glob.glob(d + "/**/*_kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
ad = [i for i, r in enumerate(rows) if "adamw_ema" in r["Kernel_Name"]]
prev_end = None
for k, i in enumerate(ad[:n]):
    e = int(rows[i]["End_Timestamp"])
    if prev_end is not None:
        seg = [r for r in rows[ad[k - 1] + 1:i + 1]]
        first = int(seg[0]["Start_Timestamp"])
        busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg if "stdadk" in r["Kernel_Name"])
        names = " ".join(f"{r['Kernel_Name'].split('(')[0].replace('void ', '').replace('stdadk::', '')[:12]}:"
                         f"{(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:.0f}" for r in seg)
        print(f"step {k:3d} wall {(e - prev_end) / 1e3:7.1f} us  idle before first kernel {(first - prev_end) / 1e3:7.1f}  "
              f"stdadk kernels {busy / 1e3:6.1f} | {names}")
    prev_end = e
