#!/bin/bash
# One profiling pass of the bench step on the MI355X box, written under gpurun_out/<tag>/ (copy what is to be
# judged into profiles/).  rocprofv3: kernel trace + stats in one run; the PMC passes each in a run of their own
# with --kernel-trace only (FETCH_SIZE and WRITE_SIZE do not fit one pass; MI355X_MICROARCH.md).
#   usage: tools/profile_round.sh <tag> <commit> [bench args...]        (default bench args: C2, B = 4096, fp32)
set -uo pipefail
cd "$(dirname "$0")/.."
TAG=${1:-r02}; COMMIT=${2:-unknown}; shift 2 || true
ARGS=("$@")
WL=c2; BATCH=4096; DT=f32
for ((i = 0; i < ${#ARGS[@]}; i++)); do
  case "${ARGS[$i]}" in --workload) WL=${ARGS[$((i + 1))]};; --batch) BATCH=${ARGS[$((i + 1))]};; --dtype) DT=${ARGS[$((i + 1))]};; esac
done
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
BENCH=(python3 bench.py --no-cpu-baseline --no-sweep --no-pipeline --clock-warmup-ms 0 --windows 1 --steps 30 --warmup 5 "${ARGS[@]}")
CMD="rocprofv3 ... -- ${BENCH[*]}"
echo "== kernel trace + stats"; rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- "${BENCH[@]}" > "$OUT/stats.log" 2>&1 || exit 1
python tools/prof_summary.py "$OUT/stats" 45 > "$OUT/${TAG}_kernel_stats.md" || exit 1
cp "$(ls "$OUT"/stats/*/*_kernel_stats.csv | head -1)" "$OUT/${TAG}_kernel_stats.csv"
echo "== PMC FETCH_SIZE"; rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- "${BENCH[@]}" > "$OUT/fetch.log" 2>&1 || exit 1
echo "== PMC WRITE_SIZE"; rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- "${BENCH[@]}" > "$OUT/write.log" 2>&1 || exit 1
python tools/pmc_summary.py "$OUT/fetch" "$OUT/write" "$OUT/${TAG}_pmc_hbm_traffic.md" "$OUT/pmc_traffic.json" \
  "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE --output-format csv -- ${BENCH[*]}" "$WL" "$BATCH" "$DT" "$COMMIT" > /dev/null || exit 1
echo "== PMC SQ counters"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE \
  --output-format csv -d "$OUT/sq" -- "${BENCH[@]}" > "$OUT/sq.log" 2>&1 || exit 1
python tools/sq_summary.py "$OUT/sq" "$OUT/${TAG}_sq_counters.md" "rocprofv3 --kernel-trace --pmc SQ_... GRBM_GUI_ACTIVE --output-format csv -- ${BENCH[*]}" > /dev/null || exit 1
rm -rf "$OUT/stats" "$OUT/fetch" "$OUT/write" "$OUT/sq"       # the raw traces are large; the summaries stay
ls -la "$OUT"
