"""CPU: the library's HOST code under AddressSanitizer + UndefinedBehaviorSanitizer (`make -C lambdaworks_kzg_amd/csrc hostasan`:
host instrumented, device code built as always; GPU sanitizers are not available on this pool). Everything of the C ABI that
runs without a GPU -- struct layouts, the optimal-ate pairing product and G2 decompression, the SHA-256 host path, point sums,
the bounds walk of the transform's arithmetic, every entry point's error path -- is re-run through that build in a child
process; any report aborts it. The same build runs the whole GPU suite on the box (tools/host_asan_gpu.sh,
profiles/r03_host_asan_gpu_log.txt). The sanitizer build takes 2.5 minutes, so this test uses it when it is there
(LWKZG_BUILD_HOST_ASAN=1 builds it first) and skips otherwise."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "lambdaworks_kzg_amd", "lib_hostasan", "liblambdaworks_kzg.so")


def test_cpu_side_of_the_c_abi_under_address_and_ub_sanitizers():
    if os.environ.get("LWKZG_BUILD_HOST_ASAN"):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "lambdaworks_kzg_amd", "csrc"), "-j8", "hostasan"])
    if not os.path.exists(LIB):
        pytest.skip("host-sanitizer build absent: make -C lambdaworks_kzg_amd/csrc hostasan (or LWKZG_BUILD_HOST_ASAN=1)")
    src_dir = os.path.join(ROOT, "lambdaworks_kzg_amd", "csrc")
    newest = max(os.path.getmtime(os.path.join(src_dir, f)) for f in os.listdir(src_dir) if f.endswith((".hip", ".h", ".cuh", ".inc")))
    newest = max(newest, os.path.getmtime(os.path.join(ROOT, "include", "lambdaworks_kzg_amd.h")))
    if os.path.getmtime(LIB) < newest:   # a build of older sources (it is not tracked): it would be testing another library
        pytest.skip("host-sanitizer build is older than csrc/: rebuild it (make -C lambdaworks_kzg_amd/csrc hostasan, or LWKZG_BUILD_HOST_ASAN=1)")
    rt = subprocess.check_output(["/opt/rocm/lib/llvm/bin/clang", "-print-file-name=libclang_rt.asan-x86_64.so"]).decode().strip()
    if not os.path.exists(rt):
        import glob
        rt = (glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so") + [""])[0]
    assert os.path.exists(rt), "clang's ASan runtime not found"
    env = dict(os.environ, LWKZG_LIBRARY=LIB, LD_PRELOAD=rt, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_capi_cpu.py"), "-q", "-x", "-p", "no:cacheprovider"],
                         env=env, capture_output=True, timeout=1200)
    text = out.stdout.decode() + out.stderr.decode()
    assert out.returncode == 0, text[-3000:]
    assert "passed" in text and "AddressSanitizer" not in text and "runtime error" not in text, text[-3000:]
