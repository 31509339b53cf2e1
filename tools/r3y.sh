#!/bin/bash
set -uo pipefail
cd "$(dirname "$0")/.."
O=gpurun_out/r3y; mkdir -p $O
python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo rc=$?
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r3y/bench_default.json").read().strip().splitlines()[-1])
v=d["variants"]
print(round(d["value"]/1e6,2), {k: (round(x["obs_per_s"]/1e6,2), x.get("ms_per_step_windows")) for k,x in v.items() if isinstance(x,dict) and "obs_per_s" in x}, {k: round(x["obs_per_s"]/1e6,1) for k,x in d["batch_sweep"].items()})
PY
