#!/bin/bash
# final checks of the round on the MI355X box: build + smoke, GPU suite, default bench, two-rank rehearsal over gloo
set -uo pipefail
cd "$(dirname "$0")/.."
O=gpurun_out/r3x; mkdir -p $O
python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -1
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/tests.log 2>&1; tail -2 $O/tests.log
python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "default rc=$?"
STNF_BENCH_BACKEND=gloo STNF_BENCH_ONE_GPU=1 timeout -k 10 600 python bench.py --gpus 2 --steps 20 --warmup 5 --windows 2 > $O/bench_2rank_gloo.json 2> $O/bench_2rank_gloo.err; echo "2rank rc=$?"
STNF_BENCH_BACKEND=gloo STNF_BENCH_ONE_GPU=1 timeout -k 10 600 python bench.py --gpus 2 --steps 20 --warmup 5 --windows 2 --dp-mode allreduce > $O/bench_2rank_gloo_ar.json 2> $O/bench_2rank_gloo_ar.err; echo "2rank allreduce rc=$?"
python - <<'PY'
import json
for f in ("bench_default","bench_2rank_gloo","bench_2rank_gloo_ar"):
    try:
        d=json.loads(open(f"gpurun_out/r3x/{f}.json").read().strip().splitlines()[-1])
        print(f, d["n_gpus"], round(d["value"]/1e6,2), "M obs/s", round(d["ms_per_step"]*1e3,1), "us | roofline", d["roofline"]["kernel"][:22], round(d["roofline"]["frac"],3), d["roofline"].get("traffic"), "| extras", [k for k in d if k.endswith("_line") or k.startswith("inference") or k=="dp_mode_fallback"], d.get("collectives",{}).get("dp_mode"))
    except Exception as e: print(f, "ERR", e)
PY
