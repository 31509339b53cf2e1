#!/bin/bash
set -uo pipefail
cd "$(dirname "$0")/.."
O=gpurun_out/r3b; mkdir -p $O
python -m pytest tests -m gpu -q > $O/tests.log 2>&1; tail -12 $O/tests.log
bash tools/diag/grad_err_variants.sh 20000 > $O/grad_err_variants.log 2>&1; cat $O/grad_err_variants.log
STNF_BENCH_BACKEND=gloo STNF_BENCH_ONE_GPU=1 timeout -k 10 400 python bench.py --gpus 2 --steps 5 --warmup 2 --windows 3 > $O/bench_2rank_shard.json 2> $O/bench_2rank_shard.err; echo "rc=$?"; tail -3 $O/bench_2rank_shard.err; head -c 1500 $O/bench_2rank_shard.json
timeout -k 10 500 python bench.py --steps 20 --warmup 5 > $O/bench_driver_args.json 2> $O/bench_driver_args.err; echo "rc=$?"; tail -3 $O/bench_driver_args.err; head -c 900 $O/bench_driver_args.json
