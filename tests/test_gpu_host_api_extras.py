"""GPU: the additive host-side entry points of round 3 -- lwkzg_runtime_init, lwkzg_timing_report, lwkzg_reserve_streams --
behave as include/lambdaworks_kzg_amd.h says (nothing the reference has a counterpart for: its load is
/root/reference/src/lib.rs:709-776, single-threaded, with no device state to report on)."""
import pytest

import blobs as B
from conftest import R, SETUP_PATH, tau_closed_form

pytestmark = pytest.mark.gpu


def test_timing_report_of_a_load_and_a_table_build(K, oracle):
    from lambdaworks_kzg_amd import capi
    capi.runtime_init()                                        # idempotent; never launches on the NULL stream
    ts = K.TrustedSetup.from_file(SETUP_PATH)
    try:
        rep = ts.timing_report()
        load, build = rep["load"], rep["last_table_build"]
        assert load["total_ms"] > 0 and load["points_and_tables_ms"] > 0
        assert abs(load["total_ms"] - (load["context_ms"] + load["points_and_tables_ms"] + load["g2_and_fft_ms"] + load["default_table_ms"])) < 5.0
        assert build["bits"] == ts.direct_table_bits() and build["row_bytes"] == ts.direct_row_bytes()
        assert build["table_bytes"] == capi.direct_table_bytes(build["bits"], build["row_bytes"])
        assert build["kernels_ms"] > 0 and build["total_ms"] <= load["default_table_ms"] + 1.0
        ts.enable_direct_table(11)
        b2 = ts.timing_report()["last_table_build"]
        assert b2["bits"] == 11 and b2["free_old_ms"] >= 0 and b2["table_malloc_ms"] >= 0 and b2["kernels_ms"] > 0
        assert b2["total_ms"] >= b2["table_malloc_ms"] + b2["kernels_ms"]
        blob = B.synthetic_blob(41000)
        assert K.blob_to_kzg_commitment(blob, ts) == tau_closed_form(oracle, B.blob_scalars(blob))
    finally:
        ts.free()


def test_reserve_streams_creates_the_second_context_up_front(K, oracle):
    """after lwkzg_reserve_streams(s, n, 2) two calls on two caller streams run without a first-use allocation, and give the
    bytes of the one-stream calls"""
    import torch
    ts = K.TrustedSetup.from_file(SETUP_PATH)
    try:
        n = 96
        ts.reserve(n, caller_streams=2)
        free_before = torch.cuda.mem_get_info()[0]
        data = B.synthetic_batch(42000, n)
        d_blobs = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
        streams = [torch.cuda.Stream() for _ in range(2)]
        outs = [torch.empty(48 * n, dtype=torch.uint8, device="cuda") for _ in range(2)]
        stats = [torch.full((n,), 9, dtype=torch.int32, device="cuda") for _ in range(2)]
        torch.cuda.synchronize()
        free_before = torch.cuda.mem_get_info()[0]
        for rep in range(3):
            for k in range(2):
                K.blob_to_kzg_commitment_batch_device(outs[k].data_ptr(), d_blobs.data_ptr(), n, ts, streams[k].cuda_stream, stats[k].data_ptr())
        torch.cuda.synchronize()
        # no workspace was allocated by the overlapped calls (a second one is 1.8 MB per blob = 170 MB here; the runtime's own
        # first-launch bookkeeping on the two new streams is a few tens of MB)
        assert torch.cuda.mem_get_info()[0] >= free_before - (96 << 20)
        assert all(int(s.abs().sum()) == 0 for s in stats) and torch.equal(outs[0], outs[1])
        got = bytes(outs[0].cpu().numpy().tobytes())
        for i in (0, n // 2, n - 1):
            assert got[48 * i:48 * i + 48] == tau_closed_form(oracle, B.blob_scalars(data[i * B.BYTES_PER_BLOB:(i + 1) * B.BYTES_PER_BLOB]))
    finally:
        ts.free()


def test_default_engine_exactly_one_colliding_lane_per_blob(K, oracle):
    """ADVICE r03, on the engine a plain load selects (the generic-plan stream of k_direct_accumulate_asm): one lane per blob meets a
    row equal / opposite to its accumulator, late in its walk; the redo pass must put every such blob right (closed form).
    tests/test_gpu_parity.py runs the same on the 10 / 12 / 14 / 15 / 16-bit tables."""
    import random
    from test_gpu_parity import _one_colliding_lane_blob
    ts = K.TrustedSetup.from_file(SETUP_PATH)
    try:
        c = ts.direct_table_bits()
        assert c in (10, 11, 12, 13)
        rnd = random.Random(4099)
        sets = [_one_colliding_lane_blob(rnd, c, negate=(b % 2 == 1)) for b in range(512)]
        data = b"".join(b"".join(x.to_bytes(32, "big") for x in ss) for ss in sets)
        got = K.blob_to_kzg_commitment_batch(data, ts)
        bad = [b for b in range(len(sets)) if got[b] != tau_closed_form(oracle, sets[b])]
        assert not bad, (len(bad), bad[:8])
    finally:
        ts.free()
