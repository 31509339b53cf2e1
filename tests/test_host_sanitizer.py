"""Host-side sanitizer pass (CPU build container only): libstdadk's HOST code
-- descriptor validation, workspace planner, job tables, launch code -- built with AddressSanitizer + UBSan
(tools/build_asan.sh; the device code stays unsanitized, GPU sanitizers are not available on this pool) and driven with
STDADK_DRY_RUN=1 (no HIP call) through the package's own host code by tools/asan_driver.py."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLANG = "/opt/rocm/lib/llvm/bin/clang"


@pytest.mark.skipif(not (os.path.exists(CLANG) and shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")),
                    reason="needs the ROCm toolchain of the build container")
def test_host_code_is_clean_under_asan_and_ubsan():
    subprocess.run(["bash", os.path.join(ROOT, "tools", "build_asan.sh")], check=True, timeout=1200,
                   stdout=subprocess.DEVNULL)
    rt = subprocess.run([CLANG, "-print-file-name=libclang_rt.asan-x86_64.so"], check=True, capture_output=True,
                        text=True).stdout.strip()
    assert os.path.exists(rt), rt
    env = dict(os.environ, LD_PRELOAD=rt, STDADK_DRY_RUN="1",
               STDADK_LIB=os.path.join(ROOT, "st-dadk_amd", "lib", "libstdadk_asan.so"),
               # the interpreter's own allocations are not ours to judge: no leak report at exit
               ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:abort_on_error=0:exitcode=23",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "asan_driver.py")], env=env, capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, (r.returncode, r.stdout[-2000:], r.stderr[-4000:])
    assert "asan driver:" in r.stdout and "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-4000:]
