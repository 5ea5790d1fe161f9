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
