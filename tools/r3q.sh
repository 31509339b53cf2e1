#!/bin/bash
set -uo pipefail
cd "$(dirname "$0")/.."
O=gpurun_out/r3q; mkdir -p $O
python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -2
python bench.py --workload c4 --no-cpu-baseline --no-sweep > $O/bench_c4.json 2> $O/bench_c4.err; echo "c4 rc=$?"
python bench.py --workload c3 --no-cpu-baseline > $O/bench_c3.json 2> $O/bench_c3.err; echo "c3 rc=$?"
python bench.py --workload c4 --batch 16384 --no-cpu-baseline --no-sweep > $O/bench_c4_b16384.json 2> $O/bench_c4_b16384.err; echo "c4 16384 rc=$?"
python - <<'PY'
import json
for f in ("bench_c4","bench_c3","bench_c4_b16384"):
    d=json.loads(open(f"gpurun_out/r3q/{f}.json").read().strip().splitlines()[-1])
    print(f, round(d["value"]/1e6,2), "M obs/s", round(d["ms_per_step"]*1e3,1), "us | roofline", d["roofline"]["kernel"][:22], round(d["roofline"]["frac"],3), "| kernels", {k[:18]:v for k,v in list(d["kernels_us_per_step"].items())[:5]}, "| rbf", round(d["rbf_build"]["frac"],2), round(d["rbf_build_past_l3"]["frac"],2), "| bf16", {k: round(v["obs_per_s"]/1e6,1) for k,v in d.get("bf16_mlp",{}).items() if isinstance(v,dict)})
PY
