"""AddressSanitizer + UndefinedBehaviorSanitizer over the CPU builds (SURVEY
section 5's sanitizer row; GPU ASan is not available on this pool): the fp64
checker (gcc) and the host build of the product's `__host__ __device__`
arithmetic (tests/host_harness.hip: the one-, and eight-lane substeps, both
contact models, the sweeps and the active-set solve on captured systems), each
loaded into a python started with the sanitizer runtime preloaded and driven
through tests/sanitizer_workload.py. A report of either sanitizer fails the run
(-fno-sanitize-recover, halt_on_error)."""

import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKLOAD = os.path.join(ROOT, "tests", "sanitizer_workload.py")
SAN_ENV = {"ASAN_OPTIONS": "detect_leaks=0:halt_on_error=1:abort_on_error=0", "UBSAN_OPTIONS": "halt_on_error=1:print_stacktrace=1", "OMP_NUM_THREADS": "4"}


def run(env, *args):
    full = dict(os.environ)
    full.update(SAN_ENV)
    full.update(env)
    out = subprocess.run([sys.executable, WORKLOAD, *args], capture_output=True, text=True, timeout=1500, env=full, cwd=ROOT)
    report = out.stdout[-2000:] + out.stderr[-4000:]
    assert out.returncode == 0 and "runtime error" not in out.stderr and "AddressSanitizer" not in out.stderr, report
    return out.stdout


def test_oracle_under_address_and_undefined_behaviour_sanitizers():
    runtime = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(runtime) or not os.path.exists(runtime):
        pytest.skip("gcc's libasan is not installed")
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "sanitize"], check=True, capture_output=True)
    lib = os.path.join(ROOT, "oracle", "_build", "libupkie_oracle_asan.so")
    assert "oracle workload done" in run({"LD_PRELOAD": runtime, "UPKIE_ORACLE_LIBRARY": lib}, "oracle")


def test_device_arithmetic_on_host_under_address_and_undefined_behaviour_sanitizers():
    if shutil.which("hipcc") is None:
        pytest.skip("hipcc not available")
    clang = "/opt/rocm/lib/llvm/bin/clang"
    runtime = subprocess.run([clang, "-print-file-name=libclang_rt.asan-x86_64.so"], capture_output=True, text=True).stdout.strip() if os.path.exists(clang) else ""
    if not os.path.isabs(runtime) or not os.path.exists(runtime):
        pytest.skip("clang's AddressSanitizer runtime is not installed")
    lib = os.path.join(ROOT, "tests", "_host_harness_asan.so")
    src = os.path.join(ROOT, "tests", "host_harness.hip")
    csrc = os.path.join(ROOT, "upkie_amd", "csrc")
    deps = [src] + [os.path.join(csrc, n) for n in os.listdir(csrc)]
    if not os.path.exists(lib) or any(os.path.getmtime(d) > os.path.getmtime(lib) for d in deps):
        # the sanitizers instrument the HOST pass (the CPU build of the arithmetic); the device pass is compiled plain (-fno-gpu-sanitize:
        # the module constructor wants its fat binary, and GPU ASan is not available on this pool)
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=address,undefined", "-fno-gpu-sanitize",
                        "-fno-sanitize-recover=undefined", "-shared-libsan", "-std=c++17", "-shared", "-fPIC", src, "-o", lib], check=True, capture_output=True)
    assert "harness workload done" in run({"LD_PRELOAD": runtime}, "harness", lib)
