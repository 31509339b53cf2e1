#!/bin/bash
# BUILD CONTAINER ONLY (the reference never travels to the GPU box): runs the reference's OWN test files and its
# OWN training driver, unmodified, against this package's `stnf` (put first on PYTHONPATH), on host tensors --
# SURVEY.md 7 step 2's gate for the drop-in boundary.  Nothing is written under /root/reference (no bytecode, no
# pytest cache; the driver's results/ directory is created relative to the working directory, a scratch dir).
#   usage: tools/run_reference_suite.sh [scratch dir]      -> prints a summary, exit code 0 iff everything passed
set -uo pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
REF=/root/reference
WORK="${1:-/tmp/stdadk_refrun}"
[ -d "$REF" ] || { echo "no $REF here: this script only runs in the build container"; exit 2; }
mkdir -p "$WORK" && cd "$WORK"
export PYTHONDONTWRITEBYTECODE=1
export PYTHONPATH="$ROOT/st-dadk_amd:$REF"
python - <<'PY' || exit 1
import stnf, os
assert os.path.realpath(stnf.__file__).startswith(os.path.realpath(os.environ["PYTHONPATH"].split(":")[0])), stnf.__file__
print("stnf resolves to", stnf.__file__)
PY
echo "== the reference's tests (tests/stnf/models/*.py), importlib import mode so that pytest leaves sys.path alone"
python -m pytest "$REF/tests/stnf/models" -q -p no:cacheprovider --import-mode=importlib 2>&1 | tail -4 | tee tests.log
t_rc=${PIPESTATUS[0]}
echo "== the reference's driver (scripts/train_st_interp.py, unmodified), 1 epoch on data/2a/2a_8.csv, device cpu:"
echo "   (a) the SHIPPED config (GMM learnable knots, 5 quantiles)  (b) its STDK corner (uniform fixed knots, MSE)"
python - <<PY
import yaml
cfg = yaml.safe_load(open("$REF/configs/config_st_interp.yaml"))
cfg.update(data_file="$REF/data/2a/2a_8.csv", epochs=1, n_experiments=1, device="cpu", warmup_epochs=0, tag="refsuite_a")
yaml.safe_dump(cfg, open("cfg_a.yaml", "w"))
cfg.update(spatial_init_method="uniform", spatial_learnable=False, regression_type="mean", tag="refsuite_b")
yaml.safe_dump(cfg, open("cfg_b.yaml", "w"))
PY
d_rc=0
for c in a b; do
  python "$REF/scripts/train_st_interp.py" --config cfg_$c.yaml > driver_$c.log 2>&1 || d_rc=1
  if grep -q "Traceback" driver_$c.log; then d_rc=1; fi
  echo "-- config ($c):"
  grep -E "parameters|Test  - |CRPS|Traceback|Error" driver_$c.log | tail -8
done
echo "== summary: reference tests rc=$t_rc, driver rc=$d_rc"
[ "$t_rc" -eq 0 ] && [ "$d_rc" -eq 0 ]
