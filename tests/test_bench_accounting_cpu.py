"""CPU: the arithmetic behind bench.py's `roofline` object (engine_picture), on a fake kernel clock -- so that a slip in the accounting
(algorithmic bytes per launch, multiply-adds per mixed addition, the measured ceilings) shows up here and not as a wrong figure in a GPU
run. SURVEY.md 8(d) fixes the per-unit figure: 524,336 algorithmic bytes per 4096-term MSM."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


class _Lib:
    def lwkzg_direct_num_windows(self, bits):
        return -(-255 // bits)

    def lwkzg_msm_num_windows(self):
        return 20

    def lwkzg_msm_window_bits(self):
        return 13


class _K:
    def lib(self):
        return _Lib()


def test_headline_roofline_object_from_a_known_kernel_clock():
    b = _bench()
    steps, n, ms = 10, 1024, 9.03
    prof = {"k_direct_accumulate_asm": {"launches": steps, "total_ms": ms * steps}, "k_direct_fold_lanes": {"launches": steps, "total_ms": 2.1}}
    r, nwin = b.engine_picture(_K(), None, 16, prof, elapsed=9.43e-3 * steps, steps=steps, n=n)
    assert nwin == 16 and r["kernel"] == "k_direct_accumulate_asm" and r["launches_per_step"] == 1
    for key in ("bound", "achieved", "peak", "unit", "frac"):           # the contract's keys (`traffic` is attached by main() from the committed PMC file)
        assert key in r
    assert r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["avg_launch_ms"] - ms) < 1e-9
    assert r["algorithmic_bytes_per_launch"] == 1024 * 524336
    assert abs(r["achieved"] - 1024 * 524336 / (ms * 1e-3) / 1e9) < 1e-6 and abs(r["frac"] - r["achieved"] / 8000.0) < 1e-12
    im = r["int_mad"]
    adds = 1024 * 4096 * 16 * (1 - 2.0 ** -16)
    assert b.MADS_PER_MIXED_ADD == 3542
    assert abs(im["mad_u64_u32_per_launch"] - adds * 3542) < 1.0
    rate = adds * 3542 / (ms * 1e-3)
    assert abs(im["achieved_Gmad_per_s"] - rate / 1e9) < 1e-3
    assert abs(im["frac_of_theoretical"] - rate / (256 * 4 * 64 / 4 * 2.4e9)) < 1e-9
    # the measured ceilings: a pure multiply-add stream on random 28-bit operands (profiles/r03_ubench_sustain.txt)
    assert abs(im["peak_sustained_random_operands_Gmad_per_s"] - 65536 / 1.832e-9 / 1e9) < 1e-6
    assert 0.68 < im["frac_of_sustained_random_operands"] < 0.80
    assert im["valu_instructions_per_mixed_addition"] == 4256
    assert 0.86 < im["valu_instruction_rate_frac_of_a_pure_random_mad_stream"] < 0.92
    g = r["gather"]
    assert abs(g["rows_per_launch"] - adds) < 1.0 and g["bytes_per_launch"] == g["rows_per_launch"] * 112


def test_compiler_arm_and_bucket_engine_pick_their_own_kernel():
    b = _bench()
    prof = {"k_direct_accumulate": {"launches": 5, "total_ms": 50.5}}
    r, _ = b.engine_picture(_K(), None, 16, prof, elapsed=0.0525, steps=5, n=1024)
    assert r["kernel"] == "k_direct_accumulate" and r["int_mad"]["valu_instructions_per_mixed_addition"] == 4814
    prof = {"k_bucket_accumulate": {"launches": 10, "total_ms": 58.0}}     # two sub-batch launches per step
    r, nwin = b.engine_picture(_K(), None, 0, prof, elapsed=0.08, steps=5, n=1024)
    assert r["kernel"] == "k_bucket_accumulate" and nwin == 20 and r["launches_per_step"] == 2 and r["gather"] is None
    assert r["algorithmic_bytes_per_launch"] == 512 * 524336
    assert "valu_instructions_per_mixed_addition" not in r["int_mad"]
    prof = {"k_bucket_accumulate_asm": {"launches": 10, "total_ms": 54.0}, "k_digit_sort": {"launches": 10, "total_ms": 3.4}}   # the hand-scheduled accumulation (round 4)
    r, _ = b.engine_picture(_K(), None, 0, prof, elapsed=0.075, steps=5, n=1024)
    assert r["kernel"] == "k_bucket_accumulate_asm" and abs(r["avg_launch_ms"] - 5.4) < 1e-9 and r["int_mad"]["frac_of_theoretical"] > 0.3


# ---- the ONE stdout line (a 21 KB line lost its head in the driver's bounded tail of stdout) ------------------------------

LINE_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
             "config", "roofline", "cpu_baseline")
ROOFLINE_KEYS = ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_launch", "avg_launch_ms", "int_mad")
CPU_KEYS = ("value", "unit", "cores", "kind", "sample", "single_thread_ops_per_s", "gpu_outputs_match_oracle")


def _detail_files():
    import glob
    return sorted(glob.glob(os.path.join(ROOT, "profiles", "r0*_bench_line.json")) + glob.glob(os.path.join(ROOT, "profiles", "r0*_bench_detail.json")))


def _check_line(b, text):
    import json
    assert len(text.encode()) < b.LINE_LIMIT <= 8000, len(text)
    assert "\n" not in text
    line = json.loads(text)
    assert list(line)[:12] == list(LINE_KEYS[:12])                # the contract's fields lead the line
    for k in LINE_KEYS:
        assert k in line, k
    assert "workload" in line["config"]
    for k in ROOFLINE_KEYS:
        assert k in line["roofline"], k
    assert "frac_of_theoretical" in line["roofline"]["int_mad"]
    for k in CPU_KEYS:
        assert k in line["cpu_baseline"], k
    assert line["dtype"] in ("u32", "u32 limbs (381-bit Fp / 255-bit Fr Montgomery, integer)")
    return line


def test_stdout_line_of_every_committed_detail_file_fits_the_drivers_tail():
    """compact_line on the detail dictionaries of earlier rounds' real runs (round 3's was the 21 KB line itself)"""
    import json
    b = _bench()
    files = _detail_files()
    assert files
    for f in files:
        res = json.load(open(f))
        if "cpu_baseline" not in res or "frac_of_theoretical" not in res.get("roofline", {}).get("int_mad", {}):
            continue                                               # (round 1's line had no theoretical ceiling yet)
        line = _check_line(b, b.compact_line(res, "bench_detail.json"))
        assert line["value"] == float("%.6g" % res["value"])
        if "configs" in res:
            for name, leg in res["configs"].items():
                if isinstance(leg, dict) and "value" in leg:
                    assert set(line["configs"][name]) <= {"value", "unit", "ms_per_step", "steps", "cold_value", "kernel", "kernel_ms", "kernel_sum_over_wall", "ntt_roofline", "roofline"}
            assert "achieved" in line["configs"]["ckzg_commit_b1024_with_ntt"]["ntt_roofline"]


def test_stdout_line_worst_case_stays_under_the_limit_and_keeps_its_head():
    """a detail dictionary far fatter than any real one: thirty legs with forty kernels each, every string padded, every float at full
    precision -- the line sheds its optional tail, never the contract's fields"""
    import json
    import math
    b = _bench()
    res = json.load(open(_detail_files()[-1]))
    kern = {"k_some_kernel_with_a_long_name_%02d" % i: {"launches": 10 + i, "avg_ms": math.pi * (i + 1)} for i in range(40)}
    res["kernels"] = kern
    res.setdefault("configs", {})
    for i in range(30):
        res["configs"]["leg_%02d_with_a_long_descriptive_name" % i] = {"workload": "w" * 600, "value": math.e * 1e5, "unit": "ops/s", "steps": 10, "warmup": 3,
                                                                     "ms_per_step": math.pi, "kernels": kern, "note": "n" * 900}
    res["configs"]["broken_leg"] = {"error": "RuntimeError(" + "x" * 5000 + ")"}
    res["config"]["workload"] = res["config"]["workload"] + " " + "y" * 300
    res["cpu_baseline"]["sample"] = "s" * 2000
    res["roofline"]["traffic_source"] = "t" * 1000
    line = _check_line(b, b.compact_line(res, "bench_detail.json"))
    assert "dropped_for_size" in line and "configs" in line["dropped_for_size"]


def test_line_numbers_are_rounded_not_truncated():
    b = _bench()
    assert b._round({"a": 116634.22217, "b": [0.00799717123, 3], "c": float("nan"), "d": "s"}) == {"a": 116634.0, "b": [0.00799717, 3], "c": None, "d": "s"}


def test_launcher_free_multi_gpu_command(monkeypatch):
    """`python bench.py --gpus N` with no launcher in the environment starts torch.distributed.run itself, before anything touches the GPU"""
    b = _bench()
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 7
    import subprocess
    monkeypatch.setattr(subprocess, "call", fake_call)
    assert b.spawn_ranks(4, ["--gpus", "4", "--steps", "2"]) == 7
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-4:] == ["--gpus", "4", "--steps", "2"]
    assert cmd[cmd.index("--master-port") + 2].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert "torch" not in b.__dict__                         # bench.py imports torch inside main(), after the spawn decision
