"""The library's HOST code (plan builder, workspace layout, tensor tables, error paths) under AddressSanitizer + UBSan +
LeakSanitizer (tests/asan/: hipcc with -Xarch_host -fsanitize=address,undefined; the device code is not instrumented - GPU
sanitizers are not available on the pool).  No GPU needed: nothing is launched."""
import glob
import os
import shutil
import subprocess
from pathlib import Path

import pytest

REPO = Path(__file__).resolve().parent.parent


@pytest.mark.slow
def test_plan_builder_is_clean_under_asan_and_ubsan():
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    rt = glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so")
    if not Path(hipcc).exists() or not rt:
        pytest.skip("hipcc / the clang sanitizer runtime are not installed here")
    r = subprocess.run(["make", "-C", str(REPO / "tests" / "asan"), f"-j{min(8, os.cpu_count() or 1)}", f"HIPCC={hipcc}"],
                       capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stderr[-3000:]
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.dirname(rt[0]) + ":" + os.environ.get("LD_LIBRARY_PATH", ""),
               ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    r = subprocess.run([str(REPO / "tests" / "asan" / "build" / "plan_driver")], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and "no findings" in r.stdout and "ERROR" not in r.stderr, (r.stdout[-1000:], r.stderr[-3000:])
