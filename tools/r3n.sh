#!/bin/bash
set -uo pipefail
cd "$(dirname "$0")/.."
O=gpurun_out/r3n; mkdir -p $O
python -m pytest tests -m gpu -q 2>&1 | tail -4
python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; tail -2 $O/bench_default.err
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_args.json 2> $O/bench_driver_args.err; echo "bench rc=$?"
python - <<'PY'
import json
for f in ("bench_default","bench_driver_args"):
    d=json.loads(open(f"gpurun_out/r3n/{f}.json").read().strip().splitlines()[-1])
    print(f, round(d["value"]/1e6,2), "M obs/s", round(d["ms_per_step"]*1e3,1), "us; windows", round(d["value_windows"]["min"]/1e6,2), round(d["value_windows"]["max"]/1e6,2), "| roofline", d["roofline"]["kernel"][:20], round(d["roofline"]["frac"],3), round(d["roofline"]["avg_launch_us"],1), "us | rbf", round(d["rbf_build"]["frac"],2), round(d["rbf_build_past_l3"]["frac"],2), "| sweep", {k: round(v["obs_per_s"]/1e6,1) for k,v in d.get("batch_sweep",{}).items()}, "| bf16", {k: round(v["obs_per_s"]/1e6,1) for k,v in d.get("bf16_mlp",{}).items() if isinstance(v,dict)})
PY
