"""Per-kernel HBM traffic per launch from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE collected
separately, as MI355X_MICROARCH.md's HBM section prescribes; values are KiB per dispatch; gfx950
under-counts wide coalesced reads by 2x, hence traffic = 2*FETCH + WRITE).
usage: python tools/pmc_summary.py <fetch dir> <write dir> <out.md> <out.json> "<command line>" [workload batch dtype commit]
The json records what the figures were measured on (workload, batch, dtype, the sha of the kernel sources, the
commit): bench.py quotes a figure as `roofline.traffic` only when all of them match the run."""
import csv
import glob
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def per_kernel(d, counter):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    acc = {}
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"]
        if "stdadk::" not in name:
            continue
        short = re.sub(r"^void ", "", name).split("(")[0].replace("stdadk::", "")
        # a kernel launched at several grid sizes in one run (rbf_build: the bench batch and the past-L3 footprint)
        # is also kept per grid size, "<kernel>@<threads in the grid>"
        for key in (short, f"{short}@{r.get('Grid_Size', '?')}"):
            a = acc.setdefault(key, [0, 0.0])
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    grids = {}
    for k in acc:
        if "@" in k:
            grids.setdefault(k.split("@")[0], []).append(k)
    for base, ks in grids.items():          # one grid size only: the per-grid entry says nothing new
        if len(ks) == 1:
            del acc[ks[0]]
    return {k: (n, tot / n) for k, (n, tot) in acc.items()}


fd, wd, out_md, out_json, cmd = sys.argv[1:6]
workload, batch, dtype, commit = (sys.argv[6:10] + ["c2", "4096", "f32", "unknown"])[:4] if len(sys.argv) > 6 else ("c2", "4096", "f32", "unknown")
F, W = per_kernel(fd, "FETCH_SIZE"), per_kernel(wd, "WRITE_SIZE")
lines = ["# HBM traffic per launch (rocprofv3 PMC, separate FETCH_SIZE and WRITE_SIZE passes)", "",
         f"Command: `{cmd}`", "",
         "| kernel | launches | FETCH_SIZE KB | WRITE_SIZE KB | traffic MB (2*F+W) |", "|---|---|---|---|---|"]
kern, full = {}, {}
for k in sorted(set(F) | set(W)):
    n, f = F.get(k, (0, 0.0))
    _, w = W.get(k, (0, 0.0))
    traffic = (2 * f + w) * 1024
    short = re.sub(r"<[^@]*", "", k)
    kern[short] = traffic
    full[short] = k
    lines.append(f"| `{k}` | {n} | {f:.1f} | {w:.1f} | {traffic / 1e6:.1f} |")
open(out_md, "w").write("\n".join(lines) + "\n")
json.dump({"note": "HBM bytes per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, C2 B=4096). "
                   "traffic = 2*FETCH_SIZE (gfx950 under-counts wide coalesced reads by 2x, MI355X_MICROARCH.md "
                   "HBM section) + WRITE_SIZE, in bytes.",
           "source": {"workload": workload, "batch": int(batch), "dtype": dtype, "commit": commit,
                      "csrc_sha16": __import__("bench").csrc_hash(), "command": cmd},
           "kernel_names": full, "kernels": kern}, open(out_json, "w"), indent=1)
print("\n".join(lines))
