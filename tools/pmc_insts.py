"""Instruction mix per launch from one rocprofv3 --pmc pass (SQ_INSTS_* / SQ_ACTIVE_INST_* / SQ_BUSY_CYCLES /
GRBM_GUI_ACTIVE): which issue port a kernel's row-local phases are bound by.
usage: python tools/pmc_insts.py <pmc dir> [kernel substring]"""
import collections, csv, glob, re, sys
d = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else ""
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if "stdadk::" not in n or want not in n:
        continue
    short = re.sub(r"^void ", "", n).split("(")[0].replace("stdadk::", "")[:60]
    a = acc[short][r["Counter_Name"]]
    a[0] += 1; a[1] += float(r["Counter_Value"])
for k, v in acc.items():
    g = {c: t / n for c, (n, t) in v.items()}
    cyc = g.get("GRBM_GUI_ACTIVE", 0) / 8
    print(f"{k}: launches {next(iter(v.values()))[0]}, kernel cycles {cyc:.0f}")
    for c, val in sorted(g.items()):
        extra = ""
        if c.startswith("SQ_INSTS_") and cyc:
            extra = f"   = {val / (cyc * 256):.3f} wave-instructions per CU-cycle"
        if c.startswith("SQ_ACTIVE_INST_") and cyc:
            extra = f"   = {val / (cyc * 1024) * 100:.1f} % of SIMD-cycles"
        print(f"    {c:28s} {val:16.0f}{extra}")
