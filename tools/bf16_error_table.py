"""Writes tests/golden/bf16_achieved.json: the bf16 configuration's error against the float64 reference per case and per
tensor, measured with the kernels of this build on the MI355X (the tests then allow 2 x these figures).  Runs the
measuring tests of tests/test_gpu_bf16.py with the assertion against the table switched off and dumps what they measured.
usage (MI355X box): python tools/bf16_error_table.py"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "st-dadk_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
os.environ["STDADK_BF16_TABLE_WRITE"] = "1"
import test_gpu_bf16 as TB
for name in ["tiny9", "default227", "default227_noln", "c2_b257", "c2_b257_noln"]:
    for dense in (False, True):
        TB.test_bf16_forward_backward_matches_emulation_and_goldens(name, dense)
for B in (4096, 9000, 20000):
    TB.test_bf16_full_batches_match_emulation(B)
out = {k: {t: float(f"{v:.3e}") for t, v in d.items()} for k, d in sorted(TB._MEASURED.items())}
json.dump(out, open(TB.ACHIEVED_FILE, "w"), indent=1, sort_keys=True)
worst = max((v, k, t) for k, d in out.items() for t, v in d.items() if t not in ("y", "loss"))
print(f"wrote {TB.ACHIEVED_FILE}: {len(out)} cases; worst gradient error {worst[0]:.2e} ({worst[1]}, {worst[2]})")
