"""GPU: environment-driven opt-in of the direct table (own module: the parity module's table fixtures, up to 240 GB, are
released before this one starts a second process on the same device)."""
import pytest

import blobs as B
from conftest import SETUP_PATH

pytestmark = pytest.mark.gpu


def test_engine_selection_from_environment(K, gpu_setup):
    """LWKZG_DIRECT_BITS: how a consumer that only knows the reference's nine symbols picks the MSM engine (fresh
    processes). Unset = the library's default (a table of at most a quarter of the free memory: 13 bits here)."""
    import subprocess
    import sys
    import os
    from conftest import ROOT
    blob = B.synthetic_blob(4242)
    want = K.blob_to_kzg_commitment(blob, gpu_setup).hex()
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r); import blobs as B; import lambdaworks_kzg_amd as K; "
            "ts = K.TrustedSetup.from_file(%r); print(ts.direct_table_bits(), K.blob_to_kzg_commitment(B.synthetic_blob(4242), ts).hex())"
            % (ROOT, os.path.join(ROOT, "tests", "golden"), SETUP_PATH))
    for setting, expect in (("14", "14"), ("0", "0"), (None, str(gpu_setup.default_bits)), ("10", "10")):
        env = dict(os.environ)
        env.pop("LWKZG_DIRECT_BITS", None)
        if setting is not None:
            env["LWKZG_DIRECT_BITS"] = setting
        out = subprocess.check_output([sys.executable, "-c", code], env=env).decode().split()
        assert out[-2:] == [expect, want], (setting, out)


def test_plain_hash_kernel_still_agrees(K, gpu_setup):
    """LWKZG_HASH_PAIRS=0 keeps the one-consumer-lane-per-blob Fiat-Shamir kernel alive as a cross-check of the
    lane-pair kernel (fresh process: the choice is read once)."""
    import os
    import subprocess
    import sys
    import torch
    from conftest import ROOT
    n = 70                                      # two workgroups, the second one partly empty
    data = B.synthetic_batch(900, n)
    d_blobs = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
    d_comm = torch.empty(48 * n, dtype=torch.uint8, device="cuda")
    d_out = torch.empty(48 * n, dtype=torch.uint8, device="cuda")
    K.blob_to_kzg_commitment_batch_device(d_comm.data_ptr(), d_blobs.data_ptr(), n, gpu_setup, None, None)
    K.compute_blob_kzg_proof_batch_device(d_out.data_ptr(), d_blobs.data_ptr(), d_comm.data_ptr(), n, gpu_setup, None, None)
    torch.cuda.synchronize()
    want = bytes(d_out.cpu().numpy().tobytes()).hex()
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r); import torch, blobs as B; import lambdaworks_kzg_amd as K; "
            "ts = K.TrustedSetup.from_file(%r); n = %d; d = torch.frombuffer(bytearray(B.synthetic_batch(900, n)), dtype=torch.uint8).cuda(); "
            "c = torch.empty(48 * n, dtype=torch.uint8, device='cuda'); o = torch.empty(48 * n, dtype=torch.uint8, device='cuda'); "
            "K.blob_to_kzg_commitment_batch_device(c.data_ptr(), d.data_ptr(), n, ts, None, None); "
            "K.compute_blob_kzg_proof_batch_device(o.data_ptr(), d.data_ptr(), c.data_ptr(), n, ts, None, None); "
            "torch.cuda.synchronize(); print(bytes(o.cpu().numpy().tobytes()).hex())"
            % (ROOT, os.path.join(ROOT, "tests", "golden"), SETUP_PATH, n))
    out = subprocess.check_output([sys.executable, "-c", code], env=dict(os.environ, LWKZG_EXPERIMENTAL="1", LWKZG_HASH_PAIRS="0")).decode().split()
    assert out[-1] == want


def test_packed_and_aligned_table_rows_agree(K, gpu_setup):
    """The direct table's rows sit 128 bytes apart (one line per gather) when that table leaves headroom on the device,
    112 bytes apart (packed) otherwise; LWKZG_DIRECT_ROW forces either. Same commitments and proofs from both layouts
    (fresh processes: the choice is read once), and the loaded setup of this suite reports the layout it got."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    assert gpu_setup.direct_row_bytes() in (112, 128)
    blob = B.synthetic_blob(4343)
    comm = K.blob_to_kzg_commitment(blob, gpu_setup)
    want = [comm.hex(), K.compute_blob_kzg_proof(blob, comm, gpu_setup).hex()]
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r); import blobs as B; import lambdaworks_kzg_amd as K; "
            "ts = K.TrustedSetup.from_file(%r); b = B.synthetic_blob(4343); c = K.blob_to_kzg_commitment(b, ts); "
            "print(ts.direct_row_bytes(), c.hex(), K.compute_blob_kzg_proof(b, c, ts).hex())"
            % (ROOT, os.path.join(ROOT, "tests", "golden"), SETUP_PATH))
    for row in ("112", "128"):
        out = subprocess.check_output([sys.executable, "-c", code], env=dict(os.environ, LWKZG_DIRECT_ROW=row)).decode().split()
        assert out[-3:] == [row] + want, (row, out[-3:])


def test_ckzg_mode_from_the_environment_loads_the_lagrange_form(K, oracle, oracle_setup):
    """LWKZG_MODE=ckzg in a fresh process: what a c-kzg consumer of the nine symbols sets. The load then builds its table in the Lagrange
    form first (and the monomial one beside it when that fits), a commitment needs no transform, and every answer is the oracle's; the
    bucket engine (LWKZG_DIRECT_BITS=0) runs c-kzg commitments on the Lagrange buckets; the A/B arms of round 4's kernels
    (LWKZG_BUCKET_ASM=0, LWKZG_SORT_STAGE=0) give the same bytes."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    blob = B.synthetic_blob(4444, big_endian=False)
    rc, want_c = oracle.blob_to_kzg_commitment(blob, oracle_setup, oracle.MODE_C)
    assert rc == 0
    rc, want_p = oracle.compute_blob_kzg_proof(blob, want_c, oracle_setup, oracle.MODE_C)
    assert rc == 0
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r); import blobs as B; import lambdaworks_kzg_amd as K; "
            "ts = K.TrustedSetup.from_file(%r); b = B.synthetic_blob(4444, big_endian=False); f0 = ts.direct_table_forms(); c = K.blob_to_kzg_commitment(b, ts); "
            "print(K.get_mode(), ts.direct_table_bits(), f0, ts.direct_table_forms(), c.hex(), K.compute_blob_kzg_proof(b, c, ts).hex())"
            % (ROOT, os.path.join(ROOT, "tests", "golden"), SETUP_PATH))
    for extra, bits in (({"LWKZG_DIRECT_BITS": "12"}, "12"), ({"LWKZG_DIRECT_BITS": "0"}, "0"),
                        ({"LWKZG_DIRECT_BITS": "0", "LWKZG_BUCKET_ASM": "0", "LWKZG_SORT_STAGE": "0"}, "0")):
        env = dict(os.environ, LWKZG_MODE="ckzg", LWKZG_EXPERIMENTAL="1", **extra)
        out = subprocess.check_output([sys.executable, "-c", code], env=env).decode().split()
        mode, got_bits, forms_at_load, forms_after, c_hex, p_hex = out[-6:]
        assert mode == "1" and got_bits == bits, out[-6:]
        if bits != "0":
            assert int(forms_at_load) & 2, "the load did not build the Lagrange-form table"     # c-kzg mode was in force when the table was built
        else:
            assert forms_at_load == "0" and forms_after == "0"
        assert c_hex == want_c.hex() and p_hex == want_p.hex(), (extra, out[-6:])
