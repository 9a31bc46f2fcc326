"""CPU-side sanitizer run of the C-ABI HOST code: `make asan` builds libstylex_hip_asan.so with
AddressSanitizer + UndefinedBehaviorSanitizer on the host side of every translation unit; tests/_capi_asan_driver.py
calls the entry points with host buffers over the step's layer shapes.  Without a GPU every kernel launch fails with
"no device" — what is exercised is everything before it: argument validation, launcher eligibility (dry runs), tile /
split-K / workspace planning, weight-gradient plans, the timing bookkeeping.  (GPU-side sanitizers are not available on
this pool; the device-side guard is STYLEX_POISON=1/2.)"""
import glob
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "explaining-in-style-reproducibility-study_amd", "csrc")


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.isfile("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_capi_host_code_under_asan_ubsan():
    subprocess.check_call(["make", "-C", CSRC, "asan", "-j8"], stdout=subprocess.DEVNULL)
    lib = os.path.join(os.path.dirname(CSRC), "libstylex_hip_asan.so")
    rt = sorted(glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so"))
    assert os.path.isfile(lib) and rt, (lib, rt)
    env = dict(os.environ, LD_PRELOAD=rt[-1], STYLEX_HIP_LIB=lib,
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:halt_on_error=1:verify_asan_link_order=0",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_capi_asan_driver.py")], env=env, capture_output=True,
                       text=True, timeout=900)
    report = r.stdout[-2000:] + "\n" + r.stderr[-6000:]
    assert "AddressSanitizer" not in r.stderr and "runtime error:" not in r.stderr, report
    assert r.returncode == 0 and "asan driver ok" in r.stdout, report
