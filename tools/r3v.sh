#!/bin/bash
set -uo pipefail
cd "$(dirname "$0")/.."
bash tools/profile_round.sh r03 bed3ddb > gpurun_out/r03_profile.log 2>&1 || { tail -5 gpurun_out/r03_profile.log; exit 1; }
bash tools/profile_round.sh r03_b65536 bed3ddb --batch 65536 > gpurun_out/r03_b65536_profile.log 2>&1 || { tail -5 gpurun_out/r03_b65536_profile.log; exit 1; }
bash tools/profile_round.sh r03_bf16 bed3ddb --dtype bf16 > gpurun_out/r03_bf16_profile.log 2>&1 || { tail -5 gpurun_out/r03_bf16_profile.log; exit 1; }
mkdir -p gpurun_out/r3v
python bench.py > gpurun_out/r3v/bench_default.json 2> gpurun_out/r3v/bench_default.err; echo "default rc=$?"
python bench.py --steps 200 --warmup 20 --no-sweep > gpurun_out/r3v/bench_driver_args.json 2> gpurun_out/r3v/bench_driver_args.err; echo "driver rc=$?"
python bench.py --batch 65536 --no-cpu-baseline --no-sweep > gpurun_out/r3v/bench_b65536.json 2> gpurun_out/r3v/bench_b65536.err; echo "b65536 rc=$?"
python bench.py --workload c4 --no-cpu-baseline --no-sweep > gpurun_out/r3v/bench_c4.json 2> gpurun_out/r3v/bench_c4.err; echo "c4 rc=$?"
python - <<'PY'
import json
for f in ("bench_default","bench_driver_args","bench_b65536","bench_c4"):
    d=json.loads(open(f"gpurun_out/r3v/{f}.json").read().strip().splitlines()[-1])
    print(f, round(d["value"]/1e6,2), "M obs/s", round(d["ms_per_step"]*1e3,1), "us | roofline", d["roofline"]["kernel"][:22], round(d["roofline"]["frac"],3), d["roofline"].get("traffic"), "| cpu", d.get("cpu_baseline",{}).get("value"))
PY
