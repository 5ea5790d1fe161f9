"""ctypes loader for the CPU oracle (oracle/liboracle_kzg.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never from lambdaworks_kzg_amd/.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

OK, BADARGS, ERROR, MALLOC = 0, 1, 2, 3
MODE_R, MODE_C = 0, 1
ALGO_PIPPENGER, ALGO_NAIVE = 0, 1
BYTES_PER_BLOB = 4096 * 32


def build(force=False, native=False, out_dir=None):
    """Compile the C restatement with gcc (seconds). native=True adds -march=native and
    writes next to `out_dir` (used only by the cpu_baseline leg on the machine it times)."""
    out_dir = out_dir or _HERE
    name = "liboracle_kzg_native.so" if native else "liboracle_kzg.so"
    out = os.path.join(out_dir, name)
    srcs = [os.path.join(_HERE, f) for f in ("ref_kzg.c", "ref_g1.c", "ref_pairing.c")]
    deps = srcs + [os.path.join(_HERE, "ref_field.h")]
    if not force and os.path.exists(out) and all(os.path.getmtime(out) >= os.path.getmtime(d) for d in deps):
        return out
    flags = ["-O3", "-fPIC", "-fvisibility=hidden", "-std=gnu11", "-shared"]
    if native:
        flags.insert(1, "-march=native")
    subprocess.check_call(["gcc"] + flags + ["-o", out] + srcs)
    return out


def lib(path=None):
    global _LIB
    if path is not None:
        return _bind(C.CDLL(path))
    if _LIB is None:
        p = os.environ.get("LWKZG_ORACLE_LIBRARY")   # e.g. the AddressSanitizer build (tests/test_oracle_golden.py)
        if not p:
            p = os.path.join(_HERE, "liboracle_kzg.so")
            if not os.path.exists(p):
                build()
        _LIB = _bind(C.CDLL(p))
    return _LIB


def _bind(l):
    u8p, vp, ci, sz = C.POINTER(C.c_uint8), C.c_void_p, C.c_int, C.c_size_t
    cp = C.c_char_p
    l.orc_sha256.argtypes = [cp, cp, sz]
    l.orc_sha256.restype = None
    l.orc_load_trusted_setup_text.argtypes = [C.POINTER(vp), cp, sz, ci]
    l.orc_free_settings.argtypes = [vp]
    l.orc_free_settings.restype = None
    for f in ("orc_settings_n1", "orc_settings_n2"):
        getattr(l, f).argtypes = [vp]
    for f in ("orc_settings_g1_compressed", "orc_settings_g2_compressed"):
        getattr(l, f).argtypes = [vp]
        getattr(l, f).restype = vp
    l.orc_settings_g1_blst.argtypes = [vp, vp]
    l.orc_settings_g1_blst.restype = None
    l.orc_srs_rebuild.argtypes = [cp, ci, cp, ci]
    l.orc_blob_to_kzg_commitment.argtypes = [cp, cp, vp, ci, ci]
    l.orc_compute_kzg_proof.argtypes = [cp, cp, cp, cp, vp, ci, ci]
    l.orc_compute_blob_kzg_proof.argtypes = [cp, cp, cp, vp, ci, ci]
    l.orc_compute_challenge.argtypes = [cp, cp, cp, ci]
    l.orc_verify_kzg_proof_known_tau.argtypes = [C.POINTER(ci), cp, cp, cp, cp, C.c_uint64, ci]
    l.orc_verify_kzg_proof_known_tau_be.argtypes = [C.POINTER(ci), cp, cp, cp, cp, cp, ci]
    l.orc_msm_affine.argtypes = [cp, cp, cp, ci, ci]
    l.orc_g1_generator_mul.argtypes = [cp, cp]
    l.orc_g1_generator_mul.restype = None
    l.orc_g1_decompress_affine.argtypes = [cp, C.POINTER(ci), cp]
    l.orc_g1_compress_affine.argtypes = [cp, cp, ci]
    l.orc_fr_ntt4096.argtypes = [cp, cp, ci]
    l.orc_fr_ntt4096.restype = None
    l.orc_fp_mul_be.argtypes = [cp, cp, cp]
    l.orc_fp_mul_be.restype = None
    l.orc_fp_inv_be.argtypes = [cp, cp]
    l.orc_fp_inv_be.restype = None
    l.orc_fr_mul_be.argtypes = [cp, cp, cp]
    l.orc_fr_mul_be.restype = None
    l.orc_g1_add_affine.argtypes = [cp, C.POINTER(ci), cp, ci, cp, ci]
    l.orc_g1_mul_affine.argtypes = [cp, C.POINTER(ci), cp, cp]
    # ref_pairing.c: the plain optimal ate pairing (second opinion on the product's host pairing) and G2 helpers for its inputs
    l.orc_pairing_product_is_one.argtypes = [C.POINTER(ci), cp, cp, ci]
    l.orc_g2_generator_mul.argtypes = [cp, cp]
    l.orc_g2_generator_mul.restype = None
    l.orc_g2_compress.argtypes = [cp, cp]
    l.orc_g2_compress.restype = None
    l.orc_g2_on_curve.argtypes = [cp]
    l.orc_final_exponent.argtypes = [cp]
    l.orc_final_exponent.restype = None
    return l


class Settings:
    """Trusted setup as the oracle holds it (monomial points, reference parser rules)."""

    def __init__(self, text, check_subgroup=True, _lib=None):
        self._l = _lib or lib()
        if isinstance(text, str):
            text = text.encode()
        h = C.c_void_p()
        rc = self._l.orc_load_trusted_setup_text(C.byref(h), text, len(text), 1 if check_subgroup else 0)
        if rc != OK:
            raise ValueError("oracle: trusted setup rejected (rc=%d)" % rc)
        self.h = h
        self.n1 = self._l.orc_settings_n1(h)
        self.n2 = self._l.orc_settings_n2(h)

    @classmethod
    def from_file(cls, path, **kw):
        with open(path, "rb") as f:
            return cls(f.read(), **kw)

    def g1_compressed(self):
        return C.string_at(self._l.orc_settings_g1_compressed(self.h), 48 * self.n1)

    def g2_compressed(self):
        return C.string_at(self._l.orc_settings_g2_compressed(self.h), 96 * self.n2)

    def g1_blst(self):
        buf = C.create_string_buffer(144 * self.n1)
        self._l.orc_settings_g1_blst(self.h, buf)
        return buf.raw

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self._l.orc_free_settings(self.h)
                self.h = None
        except Exception:
            pass


def sha256(msg):
    out = C.create_string_buffer(32)
    lib().orc_sha256(out, msg, len(msg))
    return out.raw


def blob_to_kzg_commitment(blob, s, mode=MODE_R, algo=ALGO_PIPPENGER):
    assert len(blob) == BYTES_PER_BLOB
    out = C.create_string_buffer(48)
    rc = s._l.orc_blob_to_kzg_commitment(out, blob, s.h, mode, algo)
    return rc, (out.raw if rc == OK else None)


def srs_rebuild(g1_blst, g2_blst=b"", _lib=None):
    """kzgsettings_to_structured_reference_string (/root/reference/src/srs.rs:258-280), the conversion + curve checks the
    reference repeats on every API call, over the C arrays as the reference lays them out (144 bytes per blst_p1, 288 per
    blst_p2). Returns the return code."""
    assert len(g1_blst) % 144 == 0 and len(g2_blst) % 288 == 0
    return (_lib or lib()).orc_srs_rebuild(g1_blst, len(g1_blst) // 144, g2_blst, len(g2_blst) // 288)


def compute_kzg_proof(blob, z, s, mode=MODE_R, algo=ALGO_PIPPENGER):
    assert len(blob) == BYTES_PER_BLOB and len(z) == 32
    pr, y = C.create_string_buffer(48), C.create_string_buffer(32)
    rc = s._l.orc_compute_kzg_proof(pr, y, blob, z, s.h, mode, algo)
    return rc, (pr.raw if rc == OK else None), (y.raw if rc == OK else None)


def compute_blob_kzg_proof(blob, commitment, s, mode=MODE_R, algo=ALGO_PIPPENGER):
    assert len(blob) == BYTES_PER_BLOB and len(commitment) == 48
    pr = C.create_string_buffer(48)
    rc = s._l.orc_compute_blob_kzg_proof(pr, blob, commitment, s.h, mode, algo)
    return rc, (pr.raw if rc == OK else None)


def compute_challenge(blob, commitment, mode=MODE_R):
    z = C.create_string_buffer(32)
    rc = lib().orc_compute_challenge(z, blob, commitment, mode)
    return rc, (z.raw if rc == OK else None)


def verify_kzg_proof_known_tau(commitment, z, y, proof, tau=1337, mode=MODE_R):
    ok = C.c_int(0)
    rc = lib().orc_verify_kzg_proof_known_tau_be(C.byref(ok), commitment, z, y, proof, int(tau).to_bytes(32, "big"), mode)
    return rc, bool(ok.value)


def msm_affine(points_xy_be, scalars_be, algo=ALGO_PIPPENGER):
    n = len(scalars_be) // 32
    assert len(points_xy_be) == 96 * n
    out = C.create_string_buffer(48)
    rc = lib().orc_msm_affine(out, points_xy_be, scalars_be, n, algo)
    if rc != OK:
        raise ValueError("oracle msm rc=%d" % rc)
    return out.raw


def g1_generator_mul(k_int):
    out = C.create_string_buffer(48)
    lib().orc_g1_generator_mul(out, int(k_int % (1 << 256)).to_bytes(32, "big"))
    return out.raw


def g2_generator_mul(k_int):
    """[k]G2 as affine x.c0 | x.c1 | y.c0 | y.c1 (4 x 48 bytes big-endian; all zero = infinity); ref_pairing.c"""
    out = C.create_string_buffer(192)
    lib().orc_g2_generator_mul(out, int(k_int % (1 << 256)).to_bytes(32, "big"))
    return out.raw


def g2_compress(xy192):
    """ZCash compression of an affine G2 point, sign bit included"""
    out = C.create_string_buffer(96)
    lib().orc_g2_compress(out, xy192)
    return out.raw


def g2_on_curve(xy192):
    return bool(lib().orc_g2_on_curve(xy192))


def pairing_product_is_one(g1_xy96_list, g2_xy192_list):
    """prod e(P_i, Q_i) == 1 by the oracle's own pairing (affine Miller loop over Fp2[w]/(w^6 - xi), exponent (p^12-1)/r as it stands).
    Inputs: affine big-endian coordinates, all-zero = infinity. Raises on points off their curves."""
    assert len(g1_xy96_list) == len(g2_xy192_list)
    ok = C.c_int(0)
    rc = lib().orc_pairing_product_is_one(C.byref(ok), b"".join(g1_xy96_list), b"".join(g2_xy192_list), len(g1_xy96_list))
    if rc != 0:
        raise ValueError("a point is not on its curve or not canonical")
    return bool(ok.value)


def final_exponent():
    out = C.create_string_buffer(544)
    lib().orc_final_exponent(out)
    return int.from_bytes(out.raw, "big")


def g1_decompress(b48):
    xy = C.create_string_buffer(96)
    inf = C.c_int(0)
    rc = lib().orc_g1_decompress_affine(xy, C.byref(inf), b48)
    if rc != OK:
        return None
    return (xy.raw, bool(inf.value))


def g1_compress(xy96, is_inf=False):
    out = C.create_string_buffer(48)
    rc = lib().orc_g1_compress_affine(out, xy96, 1 if is_inf else 0)
    if rc != OK:
        raise ValueError("not on curve")
    return out.raw


def fr_ntt4096(data_be, inverse=False):
    out = C.create_string_buffer(BYTES_PER_BLOB)
    lib().orc_fr_ntt4096(out, data_be, 1 if inverse else 0)
    return out.raw


def fp_mul(a, b):
    out = C.create_string_buffer(48)
    lib().orc_fp_mul_be(out, a, b)
    return out.raw


def fp_inv(a):
    out = C.create_string_buffer(48)
    lib().orc_fp_inv_be(out, a)
    return out.raw


def fr_mul(a, b):
    out = C.create_string_buffer(32)
    lib().orc_fr_mul_be(out, a, b)
    return out.raw


def g1_add_affine(a_xy, a_inf, b_xy, b_inf):
    out = C.create_string_buffer(96)
    inf = C.c_int(0)
    rc = lib().orc_g1_add_affine(out, C.byref(inf), a_xy, int(a_inf), b_xy, int(b_inf))
    if rc != OK:
        raise ValueError("not on curve")
    return out.raw, bool(inf.value)


def g1_mul_affine(p_xy, k_int):
    out = C.create_string_buffer(96)
    inf = C.c_int(0)
    rc = lib().orc_g1_mul_affine(out, C.byref(inf), p_xy, int(k_int).to_bytes(32, "big"))
    if rc != OK:
        raise ValueError("not on curve")
    return out.raw, bool(inf.value)
