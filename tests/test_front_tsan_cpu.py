"""CPU: the coalescing fronts of the single-blob symbols (lambdaworks_kzg_amd/csrc/front.h -- the queues, leaders, lanes and
staging slots that engine.hip drives the GPU with) under -fsanitize=thread against a stub device: tests/front_tsan.cpp.
Many threads on one front, both modes, inputs the "device" rejects, and `run` functions that throw (std::bad_alloc in the
leader): every request gets the answer of a call of its own, the invariants hold, ThreadSanitizer reports nothing.
Reference contract: concurrent callers on one KZGSettings, /root/reference/src/lib.rs:253-283 (SURVEY 8b "Threading")."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_fronts_under_thread_sanitizer(tmp_path):
    exe = str(tmp_path / "front_tsan")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=thread", "-pthread", "-I", os.path.join(ROOT, "lambdaworks_kzg_amd", "csrc"),
                           os.path.join(ROOT, "tests", "front_tsan.cpp"), "-o", exe])
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1:exitcode=66")
    for threads, calls in ((12, 150), (32, 60)):
        out = subprocess.run([exe, str(threads), str(calls)], env=env, capture_output=True, timeout=600)
        text = out.stdout.decode() + out.stderr.decode()
        assert out.returncode == 0, text[-3000:]
        assert "ThreadSanitizer" not in text and " 0 check failures" in text, text[-3000:]
        assert "runs threw" in text and " 0 runs threw" not in text          # the exception path was really taken
