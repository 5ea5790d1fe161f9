"""CPU: pin the oracle (oracle/ref_*.c) against every golden vector and known-answer test the
reference holds for the hot path, so that the GPU parity tests can trust it."""
import hashlib

import pytest

import blobs as B
from conftest import R, TAU, hx, tau_closed_form


def test_blob_formulas_match_reference_yaml_digests():
    for name in B.BLOB_IDS:
        blob = B.make_blob(name)
        assert hashlib.sha256(blob).hexdigest() == B.BLOB_SHA256[name]


def test_sha256_matches_hashlib(oracle):
    for n in (0, 1, 55, 56, 63, 64, 65, 119, 120, 131152):
        m = (bytes(range(256)) * (n // 256 + 1))[:n]
        assert oracle.sha256(m) == hashlib.sha256(m).digest()


def test_setup_is_powers_of_tau(oracle, oracle_setup):
    g1 = oracle_setup.g1_compressed()
    # tests/lib_test.rs:271-276 asserts this hex for g1[0]
    assert g1[:48].hex() == ("97f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac58"
                             "6c55e83ff97a1aeffb3af00adb22c6bb")
    for i in (0, 1, 2, 17, 4095):
        assert oracle.g1_generator_mul(pow(TAU, i, R)) == g1[48 * i:48 * i + 48]


def test_oracle_setup_subgroup(oracle):
    # srs.rs:62: every G1 line is decompressed AND subgroup-checked; also trusted_setup_4.txt (srs.rs:282-295)
    import os
    from conftest import GOLDEN, SETUP_PATH
    s = oracle.Settings.from_file(SETUP_PATH, check_subgroup=True)
    assert (s.n1, s.n2) == (4096, 65)
    s4 = oracle.Settings.from_file(os.path.join(GOLDEN, "trusted_setup_4.txt"), check_subgroup=True)
    assert (s4.n1, s4.n2) == (4, 65)


def test_compression_kats(oracle):
    # src/compression.rs:212-221
    kat = bytes.fromhex("8d0c6eeadd3f8529d67246f77404a4ac2d9d7fd7d50cf103d3e6abb9003e5e36d8f322663ebced6707a7f46d97b7566d")
    xy, inf = oracle.g1_decompress(kat)
    assert not inf and oracle.g1_compress(xy) == kat
    # :183-189 infinity has the top two bits set; :62-75 flag handling
    assert oracle.g1_decompress(bytes([0xc0]) + bytes(47)) == (bytes(96), True)
    assert oracle.g1_decompress(bytes(48)) is None                      # not flagged compressed
    # :155-165 (0, 2) is on the curve but not in the subgroup
    bad = bytes([0x80]) + bytes(47)
    assert oracle.g1_decompress(bad) is None
    # :192-209 round trips of G and 2G
    g = oracle.g1_generator_mul(1)
    g2 = oracle.g1_generator_mul(2)
    for c in (g, g2):
        xy, inf = oracle.g1_decompress(c)
        assert oracle.g1_compress(xy) == c
    s, _ = oracle.g1_add_affine(oracle.g1_decompress(g)[0], False, oracle.g1_decompress(g)[0], False)
    assert oracle.g1_compress(s) == g2                                  # doubling branch of operate_with
    xyg = oracle.g1_decompress(g)[0]
    neg = xyg[:48] + ((0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab
                       - int.from_bytes(xyg[48:], "big")).to_bytes(48, "big"))
    assert oracle.g1_add_affine(xyg, False, neg, False)[1] is True      # P + (-P) = O


def test_lib_test_rs_behaviours_mode_r(oracle, oracle_setup):
    s = oracle_setup
    g1 = s.g1_compressed()
    one, two = (1).to_bytes(32, "big"), (2).to_bytes(32, "big")
    # tests/lib_test.rs:19-87: p(x) = 1, z = 1 -> y = 1, proof = infinity
    blob1 = one + bytes(B.BYTES_PER_BLOB - 32)
    rc, pr, y = oracle.compute_kzg_proof(blob1, one, s, oracle.MODE_R)
    assert rc == 0 and pr == bytes([0xc0]) + bytes(47) and y == one
    # tests/lib_test.rs:89-167: p(x) = x, z = 2 -> y = 2, proof = g1[0], commitment = g1[1]
    blobx = bytes(32) + one + bytes(B.BYTES_PER_BLOB - 64)
    rc, pr, y = oracle.compute_kzg_proof(blobx, two, s, oracle.MODE_R)
    assert rc == 0 and y == two and pr == g1[:48]
    rc, cm = oracle.blob_to_kzg_commitment(blobx, s, oracle.MODE_R)
    assert rc == 0 and cm == g1[48:96]
    # verification of both (closed form for the known tau stands in for the pairing)
    rc, cm1 = oracle.blob_to_kzg_commitment(blob1, s, oracle.MODE_R)
    assert oracle.verify_kzg_proof_known_tau(cm1, one, one, bytes([0xc0]) + bytes(47), TAU, oracle.MODE_R) == (0, True)
    assert oracle.verify_kzg_proof_known_tau(cm, two, two, g1[:48], TAU, oracle.MODE_R) == (0, True)
    assert oracle.verify_kzg_proof_known_tau(cm, two, one, g1[:48], TAU, oracle.MODE_R) == (0, False)


def test_g1_values_layout(oracle, oracle_setup):
    # srs.rs:131-153: canonical integers, most-significant limb first, z = 1
    import struct
    raw = oracle_setup.g1_blst()
    x = struct.unpack("<6Q", raw[:48])
    assert x[0] == 0x17f1d3a73197d794 and x[5] == 0xfb3af00adb22c6bb
    assert struct.unpack("<6Q", raw[96:144]) == (0, 0, 0, 0, 0, 1)


def test_mode_r_smoke_values_from_survey(oracle, oracle_setup):
    # SURVEY 4.3 "Mode R smoke values" (computed independently by the survey's throw-away Python)
    s = oracle_setup
    want = {"pow2": "8e5b2b903e302ad7dab80dd1726902a2b70c55f2fc78d9254863082ef02ab6b3a169133667447d0fbdfc2223d0e8e0cd",
            "r_minus_1": "8b3be5153ed301f05d1be07f998ca4c89161f367d05cfe9f19dfb1fbfca2652617d0a23a0b7d9acdd85e6a72a7ecca6e",
            "pow3": "adcd603c7f74dd55beea1be2d5d7788abcbde31210b131a3bc58939f20b384eb650d39400d1f4cb2c5f37eae125c6405"}
    for name, h in want.items():
        rc, cm = oracle.blob_to_kzg_commitment(B.make_blob(name), s, oracle.MODE_R)
        assert rc == 0 and cm.hex() == h
    blob = B.make_blob("pow2")
    cm = bytes.fromhex(want["pow2"])
    rc, z = oracle.compute_challenge(blob, cm, oracle.MODE_R)
    assert z.hex() == "197df95f3be83ab79f5a3276e5c9888514f02f896033d527b08bf732b1ee3822"
    rc, pr = oracle.compute_blob_kzg_proof(blob, cm, s, oracle.MODE_R)
    assert pr.hex() == ("b352a02445cc2f74ecf7bdb12380fd2debce3f352407514b8645d940a26079d9"
                        "b7167c68ac5957dc22c23d2aabbe471b")
    rc, pr2, y = oracle.compute_kzg_proof(blob, z, s, oracle.MODE_R)
    assert pr2 == pr and y.hex() == "516fe0a1f56f06742a8ff10b1a19d6f06f95842cc4ab0278fc1f16e92598cd7d"


def test_tau_closed_form_and_pippenger_vs_naive(oracle, oracle_setup):
    s = oracle_setup
    blob = B.synthetic_blob(0)
    rc, cm = oracle.blob_to_kzg_commitment(blob, s, oracle.MODE_R, oracle.ALGO_PIPPENGER)
    assert rc == 0 and cm == tau_closed_form(oracle, B.blob_scalars(blob))
    # short polynomial: the two MSM algorithms of the oracle agree with each other
    short = B.synthetic_blob(1)[:32 * 40] + bytes(B.BYTES_PER_BLOB - 32 * 40)
    a = oracle.blob_to_kzg_commitment(short, s, oracle.MODE_R, oracle.ALGO_PIPPENGER)
    b = oracle.blob_to_kzg_commitment(short, s, oracle.MODE_R, oracle.ALGO_NAIVE)
    assert a == b and a[1] == tau_closed_form(oracle, B.blob_scalars(short))


def test_mode_r_reduces_noncanonical_scalars(oracle, oracle_setup):
    # from_bytes_be is believed to reduce: r + 5 behaves as 5
    blob = (R + 5).to_bytes(32, "big") + bytes(B.BYTES_PER_BLOB - 32)
    rc, cm = oracle.blob_to_kzg_commitment(blob, oracle_setup, oracle.MODE_R)
    assert rc == 0 and cm == oracle.g1_generator_mul(5)


# ---- the c-kzg-4844 YAML vectors (mode C) ------------------------------------------------------

def _cases(vectors, suite):
    return vectors["suites"][suite]


def test_ckzg_blob_to_kzg_commitment(oracle, oracle_setup, vectors):
    n = 0
    for c in _cases(vectors, "blob_to_kzg_commitment"):
        blob = B.make_blob(c["input"]["blob"])
        if len(blob) != B.BYTES_PER_BLOB:
            assert c["output"] is None      # wrong length: not expressible through the C ABI
            continue
        rc, out = oracle.blob_to_kzg_commitment(blob, oracle_setup, oracle.MODE_C)
        if c["output"] is None:
            assert rc == oracle.BADARGS
        else:
            assert rc == 0 and out == hx(c["output"])
        n += 1
    assert n == 8


def test_ckzg_compute_kzg_proof(oracle, oracle_setup, vectors):
    n = 0
    for c in _cases(vectors, "compute_kzg_proof"):
        blob, z = B.make_blob(c["input"]["blob"]), hx(c["input"]["z"])
        if len(blob) != B.BYTES_PER_BLOB or len(z) != 32:
            assert c["output"] is None
            continue
        rc, pr, y = oracle.compute_kzg_proof(blob, z, oracle_setup, oracle.MODE_C)
        if c["output"] is None:
            assert rc == oracle.BADARGS
        else:
            assert rc == 0 and pr == hx(c["output"][0]) and y == hx(c["output"][1])
        n += 1
    assert n == 42


def test_ckzg_compute_blob_kzg_proof(oracle, oracle_setup, vectors):
    n = 0
    for c in _cases(vectors, "compute_blob_kzg_proof"):
        blob, cm = B.make_blob(c["input"]["blob"]), hx(c["input"]["commitment"])
        if len(blob) != B.BYTES_PER_BLOB or len(cm) != 48:
            assert c["output"] is None
            continue
        rc, pr = oracle.compute_blob_kzg_proof(blob, cm, oracle_setup, oracle.MODE_C)
        if c["output"] is None:
            assert rc == oracle.BADARGS
        else:
            assert rc == 0 and pr == hx(c["output"])
        n += 1
    assert n == 10


def test_ckzg_verify_kzg_proof_closed_form(oracle, vectors):
    n = 0
    for c in _cases(vectors, "verify_kzg_proof"):
        i = c["input"]
        cm, z, y, pr = hx(i["commitment"]), hx(i["z"]), hx(i["y"]), hx(i["proof"])
        if (len(cm), len(z), len(y), len(pr)) != (48, 32, 32, 48):
            assert c["output"] is None
            continue
        rc, ok = oracle.verify_kzg_proof_known_tau(cm, z, y, pr, TAU, oracle.MODE_C)
        if c["output"] is None:
            assert rc == oracle.BADARGS
        else:
            assert rc == 0 and ok == c["output"]
        n += 1
    assert n == 85


def test_ntt_roundtrip_and_definition(oracle):
    import random
    rnd = random.Random(5)
    vals = [rnd.randrange(R) for _ in range(4096)]
    data = b"".join(v.to_bytes(32, "big") for v in vals)
    fwd = oracle.fr_ntt4096(data, inverse=False)
    assert oracle.fr_ntt4096(fwd, inverse=True) == data
    w = pow(7, (R - 1) // 4096, R)
    for k in (0, 1, 5, 4095):       # X_k = sum x_j w^(jk)
        want = sum(v * pow(w, j * k, R) for j, v in enumerate(vals)) % R
        assert int.from_bytes(fwd[32 * k:32 * k + 32], "big") == want


def test_srs_rebuild_restatement(oracle, oracle_setup):
    """kzgsettings_to_structured_reference_string (src/srs.rs:258-280), the per-call conversion + curve checks that
    bench.py's cpu_baseline times: accepts the setup's own blst_p1 array, rejects an off-curve point (as `?` would)"""
    g1 = oracle_setup.g1_blst()
    assert len(g1) == 4096 * 144
    assert oracle.srs_rebuild(g1) == oracle.OK
    bad = bytearray(g1)
    bad[144 * 1234 + 90] ^= 4
    assert oracle.srs_rebuild(bytes(bad)) == oracle.ERROR
    # G2: the generator of the twist in the reference's blst_p2 layout (x.c0, x.c1, y.c0, y.c1; limbs most significant first)
    g2 = [0x024aa2b2f08f0a91260805272dc51051c6e47ad4fa403b02b4510b647ae3d1770bac0326a805bbefd48056c8c121bdb8,
          0x13e02b6052719f607dacd3a088274f65596bd0d09920b61ab5da61bbdc7f5049334cf11213945d57e5ac7d055d042b7e,
          0x0ce5d527727d6e118cc9cdc6da2e351aadfd9baa8cbdd3a76d429a695160d12c923ac9cc3baca289e193548608b82801,
          0x0606c4a02ea734cc32acd2b02bc28b99cb3e287e85a763af267492ab572e99ab3f370d275cec1da1aaa9075ff05f79be]
    import struct
    def limbs(v):
        return struct.pack("<6Q", *[(v >> (64 * (5 - k))) & 0xFFFFFFFFFFFFFFFF for k in range(6)])
    p2 = b"".join(limbs(v) for v in g2) + bytes(96)
    assert oracle.srs_rebuild(g1[:144], p2) == oracle.OK
    assert oracle.srs_rebuild(g1[:144], limbs(g2[0] ^ 1) + p2[48:]) == oracle.ERROR


def test_more_setups_fixtures_and_oracle_on_them(oracle):
    """tests/golden/trusted_setup_tau2.txt and trusted_setup_unstructured.txt (make_setups.py): the committed files are what
    their generator writes (its G2 arithmetic first reproduces the 65 G2 points of the tau = 1337 setup), every point is a
    valid subgroup point, and the oracle's commitment / proof on them agree with the closed forms, with the plain MSM over
    the decompressed points and with the known-tau' verifier -- so the GPU tests on these setups compare against an oracle
    that is pinned for them too (what /root/reference/tests/lib_test.rs:68 does with a random secret)."""
    import make_setups as M
    from conftest import SETUP_TAU2_PATH, SETUP_UNSTRUCTURED_PATH, unstructured_closed_form
    from proof_cases import reference_mode_proof_closed_form
    M.self_check_against_tau_1337()
    g1_tau, g1_un, g2 = M.build_texts()
    assert open(SETUP_TAU2_PATH).read() == M.setup_text(g1_tau, g2)
    assert open(SETUP_UNSTRUCTURED_PATH).read() == M.setup_text(g1_un, g2)
    s_tau = oracle.Settings.from_file(SETUP_TAU2_PATH, check_subgroup=True)
    s_un = oracle.Settings.from_file(SETUP_UNSTRUCTURED_PATH, check_subgroup=True)
    blob = B.synthetic_blob(31)
    sc = B.blob_scalars(blob)
    rc, cm = oracle.blob_to_kzg_commitment(blob, s_tau, oracle.MODE_R)
    assert rc == 0 and cm == tau_closed_form(oracle, sc, tau=M.TAU2)
    rc, pr = oracle.compute_blob_kzg_proof(blob, cm, s_tau, oracle.MODE_R)
    assert rc == 0 and pr == reference_mode_proof_closed_form(oracle, blob, cm, tau=M.TAU2)
    z = (12345).to_bytes(32, "big")
    rc, pr, y = oracle.compute_kzg_proof(blob, z, s_tau, oracle.MODE_R)
    assert oracle.verify_kzg_proof_known_tau(cm, z, y, pr, M.TAU2, oracle.MODE_R) == (0, True)
    assert oracle.verify_kzg_proof_known_tau(cm, z, y, pr, TAU, oracle.MODE_R) == (0, False)      # not the 1337 setup
    rc, cu = oracle.blob_to_kzg_commitment(blob, s_un, oracle.MODE_R)
    assert rc == 0 and cu == unstructured_closed_form(oracle, sc) and cu != cm
    pts = b"".join(oracle.g1_decompress(p)[0] for p in g1_un)
    assert oracle.msm_affine(pts, blob, oracle.ALGO_NAIVE) == cu == oracle.msm_affine(pts, blob, oracle.ALGO_PIPPENGER)


def test_fuzz_seed_fixture_and_oracle_answers(oracle, oracle_setup):
    """tests/golden/fuzz_seeds.{json,xz} (make_fuzz_seeds.py): 115 seeds of the reference's six fuzz targets, the stored
    bytes match their digests, and the oracle reproduces every stored answer in both modes (what the GPU replay in
    tests/test_gpu_fuzz_seeds.py is compared with)"""
    import fuzz_cases as F
    index, seeds = F.load_seeds()
    assert len(seeds) == 115 and sum(1 for e, _ in seeds if e["harness_calls"]) == 65
    per_target = {t: sum(1 for e, _ in seeds if e["target"] == t) for t in F.TARGETS}
    assert per_target == {"blob_to_kzg_commitment": 36, "compute_kzg_proof": 19, "compute_blob_kzg_proof": 14, "verify_kzg_proof": 2,
                          "verify_blob_kzg_proof": 2, "verify_blob_kzg_proof_batch": 42}
    for e, data in seeds:
        if not e["harness_calls"]:
            assert data is None and "expect" not in e
            continue
        for name, mode in (("reference", oracle.MODE_R), ("ckzg", oracle.MODE_C)):
            assert F.oracle_answer(oracle, oracle_setup, e["target"], data or b"", mode) == e["expect"][name], (e["target"], e["name"], name)


def test_oracle_golden_vectors_under_address_sanitizer():
    """SURVEY section 5 (sanitizers): the oracle built with -fsanitize=address,undefined (oracle/Makefile) re-runs its
    golden-vector tests in a child process; any report aborts it. CPU only: GPU sanitizers are not available here."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["make", "-s", "-C", os.path.join(root, "oracle"), "liboracle_kzg_asan.so"])
    asan_rt = subprocess.check_output(["gcc", "-print-file-name=libasan.so"]).decode().strip()
    ubsan_rt = subprocess.check_output(["gcc", "-print-file-name=libubsan.so"]).decode().strip()
    env = dict(os.environ, LWKZG_ORACLE_LIBRARY=os.path.join(root, "oracle", "liboracle_kzg_asan.so"),
               LD_PRELOAD=asan_rt + ":" + ubsan_rt, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1")
    sel = "compression or lib_test or smoke or closed_form or noncanonical or blob_to_kzg_commitment or blob_kzg_proof or ntt or srs_rebuild or sha256"
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-p", "no:cacheprovider", "-k", sel],
                         env=env, capture_output=True, timeout=900)
    assert out.returncode == 0, out.stdout.decode()[-3000:] + out.stderr.decode()[-3000:]
    assert b"passed" in out.stdout and b"AddressSanitizer" not in out.stderr and b"runtime error" not in out.stderr
