"""CPU: the library's HOST threading under ThreadSanitizer (`make -C lambdaworks_kzg_amd/csrc hosttsan`: host code instrumented, device code
built as always). What runs without a GPU and uses the library's own threads -- the parked side workers behind SideTask (the pairing's two
Miller loops, the two decompressions and the two scalar products of verify_kzg_proof), the host pool's jobs with their selective wake-ups
(Fiat-Shamir digests) -- is re-run through that build in a child process; any report fails the test. Like the ASan build it takes minutes
to compile, so the test uses it when it is there and current (LWKZG_BUILD_HOST_TSAN=1 builds it first) and skips otherwise."""
import glob
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "lambdaworks_kzg_amd", "lib_hosttsan", "liblambdaworks_kzg.so")


def test_host_threads_of_the_c_abi_under_thread_sanitizer():
    if os.environ.get("LWKZG_BUILD_HOST_TSAN"):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "lambdaworks_kzg_amd", "csrc"), "-j8", "hosttsan"])
    if not os.path.exists(LIB):
        pytest.skip("host-TSan build absent: make -C lambdaworks_kzg_amd/csrc hosttsan (or LWKZG_BUILD_HOST_TSAN=1)")
    src_dir = os.path.join(ROOT, "lambdaworks_kzg_amd", "csrc")
    newest = max(os.path.getmtime(os.path.join(src_dir, f)) for f in os.listdir(src_dir) if f.endswith((".hip", ".h", ".cuh", ".inc", ".S")))
    if os.path.getmtime(LIB) < newest:
        pytest.skip("host-TSan build is older than csrc/: rebuild it (make -C lambdaworks_kzg_amd/csrc hosttsan)")
    rt = (glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.tsan-x86_64.so") + [""])[0]
    assert os.path.exists(rt), "clang's TSan runtime not found"
    env = dict(os.environ, LWKZG_LIBRARY=LIB, LD_PRELOAD=rt, TSAN_OPTIONS="halt_on_error=0 report_signal_unsafe=0 exitcode=66")
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_capi_cpu.py"), "-q", "-x", "-p", "no:cacheprovider", "-k",
                          "host_pairing_product or pairing_variants or digests or batch_challenge or scalar_products or decompression_of"],
                         env=env, capture_output=True, timeout=1800)
    text = out.stdout.decode() + out.stderr.decode()
    assert "ThreadSanitizer" not in text, text[-4000:]
    assert out.returncode == 0 and "passed" in text, text[-3000:]
