"""ctypes binding of liblambdaworks_kzg.so -- the C ABI declared in include/lambdaworks_kzg_amd.h.

This is the same stub a maintainer of a Python consumer of lambdaworks_kzg / c-kzg-4844 would write
(INTEGRATION.md shows the cgo / Rust FFI equivalents). Function names, argument order and error
behaviour mirror the reference's extern "C" surface (/root/reference/src/lib.rs:245-829).

There is NO CPU fallback: without the built HIP library the import of `lib()` raises, and without a
GPU every compute entry point returns C_KZG_ERROR.
"""
import ctypes as C
import json
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LWKZG_LIBRARY") or os.path.join(_HERE, "lib", "liblambdaworks_kzg.so")  # override: A/B builds

C_KZG_OK, C_KZG_BADARGS, C_KZG_ERROR, C_KZG_MALLOC = 0, 1, 2, 3
MODE_REFERENCE, MODE_CKZG = 0, 1
FIELD_ELEMENTS_PER_BLOB = 4096
BYTES_PER_BLOB = 4096 * 32
BYTES_PER_COMMITMENT = 48
BYTES_PER_PROOF = 48


class KZGSettings(C.Structure):
    """#[repr(C)] KZGSettings, /root/reference/src/lib.rs:206-222."""
    _fields_ = [("fs", C.c_void_p), ("g1_values", C.c_void_p), ("g2_values", C.c_void_p)]


class FFTSettings(C.Structure):
    """#[repr(C)] FFTSettings, /root/reference/src/lib.rs:173-197."""
    _fields_ = [("max_width", C.c_uint64), ("expanded_roots_of_unity", C.c_void_p),
                ("reverse_roots_of_unity", C.c_void_p), ("roots_of_unity", C.c_void_p)]


# every symbol include/lambdaworks_kzg_amd.h declares
EXPORTED_SYMBOLS = [
    "load_trusted_setup", "load_trusted_setup_file", "free_trusted_setup", "blob_to_kzg_commitment",
    "compute_kzg_proof", "compute_blob_kzg_proof", "verify_kzg_proof", "verify_blob_kzg_proof",
    "verify_blob_kzg_proof_batch",
    "lwkzg_set_mode", "lwkzg_get_mode", "lwkzg_settings_set_mode", "lwkzg_settings_get_mode",
    "lwkzg_blob_to_kzg_commitment_batch",
    "lwkzg_compute_blob_kzg_proof_batch", "lwkzg_compute_kzg_proof_batch",
    "lwkzg_blob_to_kzg_commitment_batch_device", "lwkzg_compute_blob_kzg_proof_batch_device", "lwkzg_reserve", "lwkzg_reserve_streams",
    "lwkzg_g1_lincomb_setup_device", "lwkzg_fr_ntt4096_device",
    "lwkzg_setup_image_bytes", "lwkzg_setup_export_device", "lwkzg_setup_import_device",
    "lwkzg_device_count", "lwkzg_set_device", "lwkzg_version", "lwkzg_last_error",
    "lwkzg_profile_enable", "lwkzg_profile_reset", "lwkzg_profile_report",
    "lwkzg_msm_window_bits", "lwkzg_msm_num_windows", "lwkzg_pairing_product_is_one",
    "lwkzg_challenge_digests_host", "lwkzg_batch_challenge_host", "lwkzg_g1_msm_tiled_device", "lwkzg_g1_sum_compressed",
    "lwkzg_commit_and_prove_batch_device", "lwkzg_enable_direct_table", "lwkzg_direct_table_bits", "lwkzg_direct_table_forms", "lwkzg_enable_direct_table_forms", "lwkzg_direct_num_windows", "lwkzg_direct_row_bytes",
    "lwkzg_compute_challenges_device",
    "lwkzg_timing_report", "lwkzg_runtime_init", "lwkzg_knob_report", "lwkzg_last_proof_schedule",
    "lwkzg_multi_load", "lwkzg_multi_load_file", "lwkzg_multi_free", "lwkzg_multi_device_count", "lwkzg_multi_device", "lwkzg_multi_settings",
    "lwkzg_multi_set_mode", "lwkzg_multi_enable_direct_table", "lwkzg_multi_blob_to_kzg_commitment_batch",
    "lwkzg_multi_compute_blob_kzg_proof_batch", "lwkzg_multi_compute_kzg_proof_batch", "lwkzg_multi_verify_blob_kzg_proof_batch",
    "lwkzg_multi_g1_msm_tiled",
    "lwkzg_release_context", "lwkzg_verify_shard_begin", "lwkzg_verify_shard_partial", "lwkzg_verify_shard_free", "lwkzg_verify_shards_finish",
    "lwkzg_verify_blob_kzg_proof_batch_device", "lwkzg_verify_shard_begin_device", "lwkzg_shard_range",
    "lwkzg_multi_blob_to_kzg_commitment_batch_device", "lwkzg_multi_compute_blob_kzg_proof_batch_device",
    "lwkzg_multi_verify_blob_kzg_proof_batch_device", "lwkzg_clock_probe_mhz",
]

_lib = None


def lib():
    """Load the HIP shared library; raise loudly when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "%s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C lambdaworks_kzg_amd/csrc`). There is no CPU fallback." % LIB_PATH)
    l = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    vp, sz, ci = C.c_void_p, C.c_size_t, C.c_int
    ps = C.POINTER(KZGSettings)
    l.load_trusted_setup.argtypes = [ps, C.c_char_p, sz, C.c_char_p, sz]
    l.load_trusted_setup_file.argtypes = [ps, vp]
    l.free_trusted_setup.argtypes = [ps]
    l.blob_to_kzg_commitment.argtypes = [C.c_char_p, C.c_char_p, ps]
    l.compute_kzg_proof.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, ps]
    l.compute_blob_kzg_proof.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, ps]
    l.verify_kzg_proof.argtypes = [C.POINTER(C.c_bool), C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, ps]
    l.verify_blob_kzg_proof.argtypes = [C.POINTER(C.c_bool), C.c_char_p, C.c_char_p, C.c_char_p, ps]
    l.verify_blob_kzg_proof_batch.argtypes = [C.POINTER(C.c_bool), C.c_char_p, C.c_char_p, C.c_char_p, sz, ps]
    l.lwkzg_set_mode.argtypes = [ci]
    l.lwkzg_timing_report.argtypes = [ps, C.c_char_p, sz]
    l.lwkzg_timing_report.restype = sz
    l.lwkzg_knob_report.argtypes = [C.c_char_p, sz]
    l.lwkzg_knob_report.restype = sz
    l.lwkzg_settings_set_mode.argtypes = [ps, ci]
    l.lwkzg_settings_get_mode.argtypes = [ps]
    l.lwkzg_blob_to_kzg_commitment_batch.argtypes = [C.c_char_p, C.c_char_p, sz, ps, C.POINTER(sz)]
    l.lwkzg_compute_blob_kzg_proof_batch.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, sz, ps, C.POINTER(sz)]
    l.lwkzg_compute_kzg_proof_batch.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, sz, ps, C.POINTER(sz)]
    l.lwkzg_blob_to_kzg_commitment_batch_device.argtypes = [vp, vp, sz, ps, vp, vp]
    l.lwkzg_compute_blob_kzg_proof_batch_device.argtypes = [vp, vp, vp, sz, ps, vp, vp]
    l.lwkzg_compute_challenges_device.argtypes = [vp, vp, vp, sz, ps, vp]
    l.lwkzg_verify_blob_kzg_proof_batch_device.argtypes = [C.POINTER(C.c_bool), vp, vp, vp, sz, ps, vp]
    l.lwkzg_shard_range.argtypes = [sz, sz, sz, C.POINTER(sz), C.POINTER(sz)]
    pvp, psz = C.POINTER(vp), C.POINTER(sz)
    l.lwkzg_multi_blob_to_kzg_commitment_batch_device.argtypes = [pvp, pvp, psz, vp, psz]
    l.lwkzg_multi_compute_blob_kzg_proof_batch_device.argtypes = [pvp, pvp, pvp, psz, vp, psz]
    l.lwkzg_multi_verify_blob_kzg_proof_batch_device.argtypes = [C.POINTER(C.c_bool), pvp, pvp, pvp, psz, vp]
    l.lwkzg_verify_shard_begin_device.argtypes = [C.POINTER(vp), C.c_char_p, vp, vp, vp, sz, ps, vp]
    l.lwkzg_commit_and_prove_batch_device.argtypes = [vp, vp, vp, sz, ps, vp, vp]
    l.lwkzg_verify_shard_begin.argtypes = [C.POINTER(vp), C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, sz, ps]
    l.lwkzg_verify_shard_partial.argtypes = [C.c_char_p, vp, C.c_char_p, sz, sz]
    l.lwkzg_verify_shard_free.argtypes = [vp]
    l.lwkzg_verify_shard_free.restype = None
    l.lwkzg_verify_shards_finish.argtypes = [C.POINTER(C.c_bool), C.c_char_p, sz, sz, ps]
    l.lwkzg_release_context.argtypes = [ps]
    l.lwkzg_reserve.argtypes = [ps, sz]
    l.lwkzg_reserve_streams.argtypes = [ps, sz, ci]
    l.lwkzg_enable_direct_table.argtypes = [ps, ci]
    l.lwkzg_direct_table_bits.argtypes = [ps]
    l.lwkzg_direct_table_forms.argtypes = [ps]
    l.lwkzg_enable_direct_table_forms.argtypes = [ps, ci, ci]
    l.lwkzg_direct_num_windows.argtypes = [ci]
    l.lwkzg_direct_row_bytes.argtypes = [ps]
    l.lwkzg_g1_lincomb_setup_device.argtypes = [vp, vp, sz, ps, vp]
    l.lwkzg_fr_ntt4096_device.argtypes = [vp, vp, sz, ci, ps, vp]
    l.lwkzg_setup_image_bytes.restype = sz
    l.lwkzg_setup_export_device.argtypes = [ps, vp, vp]
    l.lwkzg_setup_import_device.argtypes = [ps, vp]
    l.lwkzg_set_device.argtypes = [ci]
    l.lwkzg_version.restype = C.c_char_p
    l.lwkzg_last_error.restype = C.c_char_p
    l.lwkzg_profile_enable.argtypes = [ci]
    l.lwkzg_profile_enable.restype = None
    l.lwkzg_profile_reset.restype = None
    l.lwkzg_profile_report.argtypes = [C.c_char_p, sz]
    l.lwkzg_profile_report.restype = sz
    l.lwkzg_pairing_product_is_one.argtypes = [C.POINTER(C.c_bool), C.c_char_p, C.c_char_p, sz]
    l.lwkzg_challenge_digests_host.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, sz]
    l.lwkzg_batch_challenge_host.argtypes = [C.c_char_p, C.c_char_p, sz, ci]
    l.lwkzg_g1_msm_tiled_device.argtypes = [vp, vp, sz, ps, vp]
    l.lwkzg_g1_sum_compressed.argtypes = [C.c_char_p, C.c_char_p, sz]
    pi = C.POINTER(C.c_int)
    l.lwkzg_multi_load.argtypes = [C.POINTER(vp), C.c_char_p, sz, C.c_char_p, sz, pi, sz]
    l.lwkzg_multi_load_file.argtypes = [C.POINTER(vp), vp, pi, sz]
    l.lwkzg_multi_free.argtypes = [vp]
    l.lwkzg_multi_free.restype = None
    l.lwkzg_multi_device_count.argtypes = [vp]
    l.lwkzg_multi_device_count.restype = sz
    l.lwkzg_multi_device.argtypes = [vp, sz]
    l.lwkzg_multi_settings.argtypes = [vp, sz]
    l.lwkzg_multi_settings.restype = ps
    l.lwkzg_multi_set_mode.argtypes = [vp, ci]
    l.lwkzg_multi_enable_direct_table.argtypes = [vp, ci]
    l.lwkzg_multi_blob_to_kzg_commitment_batch.argtypes = [C.c_char_p, C.c_char_p, sz, vp, C.POINTER(sz)]
    l.lwkzg_multi_compute_blob_kzg_proof_batch.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, sz, vp, C.POINTER(sz)]
    l.lwkzg_multi_compute_kzg_proof_batch.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, sz, vp, C.POINTER(sz)]
    l.lwkzg_multi_verify_blob_kzg_proof_batch.argtypes = [C.POINTER(C.c_bool), C.c_char_p, C.c_char_p, C.c_char_p, sz, vp]
    l.lwkzg_multi_g1_msm_tiled.argtypes = [C.c_char_p, C.c_char_p, sz, vp]
    _lib = l
    return l


class KzgError(RuntimeError):
    def __init__(self, fn, rc):
        self.rc = rc
        msg = lib().lwkzg_last_error().decode(errors="replace")
        super().__init__("%s returned %s%s" % (fn, {1: "C_KZG_BADARGS", 2: "C_KZG_ERROR", 3: "C_KZG_MALLOC"}.get(rc, rc),
                                               (": " + msg) if msg else ""))


def _check(fn, rc):
    if rc != C_KZG_OK:
        raise KzgError(fn, rc)


_libc = C.CDLL(None)
_libc.fopen.restype = C.c_void_p
_libc.fopen.argtypes = [C.c_char_p, C.c_char_p]
_libc.fclose.argtypes = [C.c_void_p]


def direct_table_bytes(window_bits, row_bytes=112):
    """HBM footprint of the direct table of that width with packed (112-byte) or line-aligned (128-byte) rows."""
    nw = lib().lwkzg_direct_num_windows(window_bits)
    if nw == 0:
        return 0
    top = 255 - window_bits * (nw - 1)
    return ((nw - 1) * 4096 * (1 << (window_bits - 1)) + 4096 * (1 << top)) * row_bytes


def shard_range(n_items, parts, k):
    """(first, count) of part k: the library's one shard rule (lwkzg_shard_range; csrc/multi.hip and dist.py both go through it)"""
    first, count = C.c_size_t(0), C.c_size_t(0)
    _check("lwkzg_shard_range", lib().lwkzg_shard_range(n_items, parts, k, C.byref(first), C.byref(count)))
    return first.value, count.value


def set_mode(mode):
    return lib().lwkzg_set_mode(mode)


def get_mode():
    return lib().lwkzg_get_mode()


def last_proof_schedule():
    """the schedule (csrc/plan.h: ProofSchedule) the last device-resident blob-proof call of this process took; -1 before the first"""
    return int(lib().lwkzg_last_proof_schedule())


def knob_report():
    """the environment knobs in effect for this process (lwkzg_knob_report; csrc/knobs.h reads them once, in one place)"""
    n = lib().lwkzg_knob_report(None, 0)
    buf = C.create_string_buffer(n)
    lib().lwkzg_knob_report(buf, n)
    return json.loads(buf.value.decode())


def clock_probe_mhz():
    """the shader clock the current device holds under a short dense multiply-add stream (lwkzg_clock_probe_mhz)"""
    mhz = C.c_double(0.0)
    lib().lwkzg_clock_probe_mhz.argtypes = [C.POINTER(C.c_double)]
    _check("lwkzg_clock_probe_mhz", lib().lwkzg_clock_probe_mhz(C.byref(mhz)))
    return mhz.value


def runtime_init():
    """first use of the HIP runtime by this process (device context, code object load): timed apart from the first real call"""
    if lib().lwkzg_runtime_init() != 0:
        raise KzgError("lwkzg_runtime_init", C_KZG_ERROR)


def set_device(ordinal):
    if lib().lwkzg_set_device(ordinal) != 0:
        raise KzgError("lwkzg_set_device", C_KZG_ERROR)


class TrustedSetup:
    """A loaded KZGSettings. Mirrors load_trusted_setup* / free_trusted_setup."""

    def __init__(self):
        self.s = KZGSettings()
        self._loaded = False

    @classmethod
    def from_file(cls, path):
        """load_trusted_setup_file(KZGSettings*, FILE*), /root/reference/src/lib.rs:779."""
        self = cls()
        fp = _libc.fopen(os.fsencode(path), b"r")
        if not fp:
            raise FileNotFoundError(path)
        try:
            _check("load_trusted_setup_file", lib().load_trusted_setup_file(C.byref(self.s), fp))
        finally:
            _libc.fclose(fp)
        self._loaded = True
        return self

    @classmethod
    def from_bytes(cls, g1_bytes, g2_bytes):
        """load_trusted_setup(out, g1_bytes, n1, g2_bytes, n2), /root/reference/src/lib.rs:709."""
        self = cls()
        _check("load_trusted_setup", lib().load_trusted_setup(C.byref(self.s), g1_bytes, len(g1_bytes) // 48,
                                                              g2_bytes, len(g2_bytes) // 96))
        self._loaded = True
        return self

    @classmethod
    def from_device_image(cls, image_dev_ptr):
        self = cls()
        _check("lwkzg_setup_import_device", lib().lwkzg_setup_import_device(C.byref(self.s), image_dev_ptr))
        self._loaded = True
        return self

    def export_device_image(self, image_dev_ptr, stream=None):
        _check("lwkzg_setup_export_device", lib().lwkzg_setup_export_device(C.byref(self.s), image_dev_ptr, stream))

    def g1_values_bytes(self):
        return C.string_at(self.s.g1_values, 4096 * 144)

    def g2_values_bytes(self):
        return C.string_at(self.s.g2_values, 65 * 288)

    def fft_settings(self):
        return FFTSettings.from_address(self.s.fs)

    def ref(self):
        return C.byref(self.s)

    def reserve(self, n, caller_streams=1):
        _check("lwkzg_reserve_streams", lib().lwkzg_reserve_streams(self.ref(), n, caller_streams))

    def set_mode(self, mode):
        """This settings object's own semantics (MODE_REFERENCE / MODE_CKZG; -1 = follow the process-wide default)."""
        prev = lib().lwkzg_settings_set_mode(self.ref(), mode)
        if prev < 0:
            raise KzgError("lwkzg_settings_set_mode", C_KZG_BADARGS)
        return prev

    def get_mode(self):
        return lib().lwkzg_settings_get_mode(self.ref())

    def enable_direct_table(self, window_bits):
        """Select the direct-table MSM of that width (10 .. 16) or the bucket engine (0); raises KzgError(C_KZG_MALLOC) if
        the table does not fit."""
        _check("lwkzg_enable_direct_table", lib().lwkzg_enable_direct_table(self.ref(), window_bits))

    def direct_table_bits(self):
        return lib().lwkzg_direct_table_bits(self.ref())

    def enable_direct_table_forms(self, window_bits, forms):
        """forms: 1 = monomial only, 2 = Lagrange only, 3 = both"""
        _check("lwkzg_enable_direct_table_forms", lib().lwkzg_enable_direct_table_forms(self.ref(), window_bits, forms))

    def direct_table_forms(self):
        """bit 0: a monomial-form direct table is live, bit 1: a Lagrange-form one (c-kzg commitments without the transform)"""
        return lib().lwkzg_direct_table_forms(self.ref())

    def direct_row_bytes(self):
        """128 = table rows aligned to 128-byte lines, 112 = packed, 0 = bucket engine."""
        return lib().lwkzg_direct_row_bytes(self.ref())

    def timing_report(self):
        """where the milliseconds of the load and of the last table build went (lwkzg_timing_report)"""
        n = lib().lwkzg_timing_report(self.ref(), None, 0)
        buf = C.create_string_buffer(n)
        lib().lwkzg_timing_report(self.ref(), buf, n)
        return json.loads(buf.value.decode())

    def free(self):
        if self._loaded:
            lib().free_trusted_setup(C.byref(self.s))
            self._loaded = False

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class _DeviceSettings:
    """the k-th device's KZGSettings of a MultiSetup, usable wherever a TrustedSetup is (the MultiSetup owns it)"""

    def __init__(self, ptr):
        self._p = ptr

    def ref(self):
        return self._p


class MultiSetup:
    """One process, several GPUs (lwkzg_multi_*, csrc/multi.hip): the setup loaded once and delivered device-to-device, batches cut
    into contiguous shards, one host thread per device, no reduction. `devices` may name a device more than once."""

    def __init__(self, h):
        self.h = h

    @classmethod
    def from_file(cls, path, devices):
        h = C.c_void_p()
        arr = (C.c_int * len(devices))(*devices)
        fp = _libc.fopen(os.fsencode(path), b"r")
        if not fp:
            raise FileNotFoundError(path)
        try:
            _check("lwkzg_multi_load_file", lib().lwkzg_multi_load_file(C.byref(h), fp, arr, len(devices)))
        finally:
            _libc.fclose(fp)
        return cls(h)

    @classmethod
    def from_bytes(cls, g1_bytes, g2_bytes, devices):
        h = C.c_void_p()
        arr = (C.c_int * len(devices))(*devices)
        _check("lwkzg_multi_load", lib().lwkzg_multi_load(C.byref(h), g1_bytes, len(g1_bytes) // 48, g2_bytes, len(g2_bytes) // 96, arr, len(devices)))
        return cls(h)

    def device_count(self):
        return lib().lwkzg_multi_device_count(self.h)

    def devices(self):
        return [lib().lwkzg_multi_device(self.h, k) for k in range(self.device_count())]

    def settings(self, k):
        p = lib().lwkzg_multi_settings(self.h, k)
        if not p:
            raise IndexError(k)
        return _DeviceSettings(p)

    def set_mode(self, mode):
        _check("lwkzg_multi_set_mode", lib().lwkzg_multi_set_mode(self.h, mode))

    def enable_direct_table(self, window_bits):
        _check("lwkzg_multi_enable_direct_table", lib().lwkzg_multi_enable_direct_table(self.h, window_bits))

    def blob_to_kzg_commitment_batch(self, blobs):
        n = len(blobs) // BYTES_PER_BLOB
        assert len(blobs) == n * BYTES_PER_BLOB
        out = C.create_string_buffer(48 * max(n, 1))
        self.first_bad = C.c_size_t(0)
        _check("lwkzg_multi_blob_to_kzg_commitment_batch", lib().lwkzg_multi_blob_to_kzg_commitment_batch(out, blobs, n, self.h, C.byref(self.first_bad)))
        raw = out.raw      # (ONE copy of the buffer: `.raw` inside the comprehension copied all 48 n bytes per element -- 6 ms of Python per 4096 results)
        return [raw[48 * i:48 * i + 48] for i in range(n)]

    def compute_blob_kzg_proof_batch(self, blobs, commitments):
        n = len(blobs) // BYTES_PER_BLOB
        assert len(blobs) == n * BYTES_PER_BLOB and len(commitments) == 48 * n
        out = C.create_string_buffer(48 * max(n, 1))
        self.first_bad = C.c_size_t(0)
        _check("lwkzg_multi_compute_blob_kzg_proof_batch",
               lib().lwkzg_multi_compute_blob_kzg_proof_batch(out, blobs, commitments, n, self.h, C.byref(self.first_bad)))
        raw = out.raw      # (ONE copy of the buffer: `.raw` inside the comprehension copied all 48 n bytes per element -- 6 ms of Python per 4096 results)
        return [raw[48 * i:48 * i + 48] for i in range(n)]

    def compute_kzg_proof_batch(self, blobs, zs):
        n = len(blobs) // BYTES_PER_BLOB
        assert len(blobs) == n * BYTES_PER_BLOB and len(zs) == 32 * n
        out, ys = C.create_string_buffer(48 * max(n, 1)), C.create_string_buffer(32 * max(n, 1))
        self.first_bad = C.c_size_t(0)
        _check("lwkzg_multi_compute_kzg_proof_batch",
               lib().lwkzg_multi_compute_kzg_proof_batch(out, ys, blobs, zs, n, self.h, C.byref(self.first_bad)))
        raw, yraw = out.raw, ys.raw
        return [(raw[48 * i:48 * i + 48], yraw[32 * i:32 * i + 32]) for i in range(n)]

    # ---- shards already in HBM: lists of per-device device pointers and counts (device k's shard on device k)
    @staticmethod
    def _ptrs(ptrs):
        return (C.c_void_p * len(ptrs))(*ptrs)

    def blob_to_kzg_commitment_batch_device(self, out_ptrs, blob_ptrs, counts):
        cnt = (C.c_size_t * len(counts))(*counts)
        self.first_bad = C.c_size_t(0)
        _check("lwkzg_multi_blob_to_kzg_commitment_batch_device",
               lib().lwkzg_multi_blob_to_kzg_commitment_batch_device(self._ptrs(out_ptrs), self._ptrs(blob_ptrs), cnt, self.h, C.byref(self.first_bad)))

    def compute_blob_kzg_proof_batch_device(self, out_ptrs, blob_ptrs, comm_ptrs, counts):
        cnt = (C.c_size_t * len(counts))(*counts)
        self.first_bad = C.c_size_t(0)
        _check("lwkzg_multi_compute_blob_kzg_proof_batch_device",
               lib().lwkzg_multi_compute_blob_kzg_proof_batch_device(self._ptrs(out_ptrs), self._ptrs(blob_ptrs), self._ptrs(comm_ptrs), cnt, self.h,
                                                                     C.byref(self.first_bad)))

    def verify_blob_kzg_proof_batch_device(self, blob_ptrs, comm_ptrs, proof_ptrs, counts):
        cnt = (C.c_size_t * len(counts))(*counts)
        ok = C.c_bool(False)
        _check("lwkzg_multi_verify_blob_kzg_proof_batch_device",
               lib().lwkzg_multi_verify_blob_kzg_proof_batch_device(C.byref(ok), self._ptrs(blob_ptrs), self._ptrs(comm_ptrs), self._ptrs(proof_ptrs), cnt,
                                                                    self.h))
        return bool(ok.value)

    def verify_blob_kzg_proof_batch(self, blobs, commitments, proofs, n):
        ok = C.c_bool(False)
        _check("lwkzg_multi_verify_blob_kzg_proof_batch",
               lib().lwkzg_multi_verify_blob_kzg_proof_batch(C.byref(ok), blobs, commitments, proofs, n, self.h))
        return bool(ok.value)

    def g1_msm_tiled(self, scalars_be):
        out = C.create_string_buffer(48)
        _check("lwkzg_multi_g1_msm_tiled", lib().lwkzg_multi_g1_msm_tiled(out, scalars_be, len(scalars_be) // 32, self.h))
        return out.raw

    def free(self):
        if self.h:
            lib().lwkzg_multi_free(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


# ---- the reference's functions, same names and argument meaning ---------------------------------

def blob_to_kzg_commitment(blob, ts):
    assert len(blob) == BYTES_PER_BLOB
    out = C.create_string_buffer(48)
    _check("blob_to_kzg_commitment", lib().blob_to_kzg_commitment(out, blob, ts.ref()))
    return out.raw


def compute_kzg_proof(blob, z_bytes, ts):
    assert len(blob) == BYTES_PER_BLOB and len(z_bytes) == 32
    proof, y = C.create_string_buffer(48), C.create_string_buffer(32)
    _check("compute_kzg_proof", lib().compute_kzg_proof(proof, y, blob, z_bytes, ts.ref()))
    return proof.raw, y.raw


def compute_blob_kzg_proof(blob, commitment_bytes, ts):
    assert len(blob) == BYTES_PER_BLOB and len(commitment_bytes) == 48
    out = C.create_string_buffer(48)
    _check("compute_blob_kzg_proof", lib().compute_blob_kzg_proof(out, blob, commitment_bytes, ts.ref()))
    return out.raw


def verify_kzg_proof(commitment_bytes, z_bytes, y_bytes, proof_bytes, ts):
    ok = C.c_bool(False)
    _check("verify_kzg_proof", lib().verify_kzg_proof(C.byref(ok), commitment_bytes, z_bytes, y_bytes, proof_bytes, ts.ref()))
    return bool(ok.value)


def verify_blob_kzg_proof(blob, commitment_bytes, proof_bytes, ts):
    ok = C.c_bool(False)
    _check("verify_blob_kzg_proof", lib().verify_blob_kzg_proof(C.byref(ok), blob, commitment_bytes, proof_bytes, ts.ref()))
    return bool(ok.value)


def verify_blob_kzg_proof_batch(blobs, commitments_bytes, proofs_bytes, n, ts):
    ok = C.c_bool(False)
    _check("verify_blob_kzg_proof_batch",
           lib().verify_blob_kzg_proof_batch(C.byref(ok), blobs, commitments_bytes, proofs_bytes, n, ts.ref()))
    return bool(ok.value)


def verify_blob_kzg_proof_batch_device(blobs_ptr, comm_ptr, proofs_ptr, n, ts, stream=None):
    """verify_blob_kzg_proof_batch on DEVICE pointers (produced on `stream`); the verdict is a host bool, the call synchronous"""
    ok = C.c_bool(False)
    _check("lwkzg_verify_blob_kzg_proof_batch_device",
           lib().lwkzg_verify_blob_kzg_proof_batch_device(C.byref(ok), blobs_ptr, comm_ptr, proofs_ptr, n, ts.ref(), stream))
    return bool(ok.value)


# ---- sharded batch verification (one batch, one r, one pairing check; include/lambdaworks_kzg_amd.h) ---------

VERIFY_RECORD_BYTES = 160
VERIFY_PARTIAL_BYTES = 328


class VerifyShard:
    """This process's shard of a sharded verify_blob_kzg_proof_batch (lwkzg_verify_shard_*)."""

    def __init__(self, blobs, commitments_bytes, proofs_bytes, n_local, ts):
        assert len(blobs) == n_local * BYTES_PER_BLOB and len(commitments_bytes) == len(proofs_bytes) == 48 * n_local
        self.n = n_local
        self.ts = ts  # the shard's device buffers belong to the setup's context: keep it alive as long as the shard
        self.h = C.c_void_p()
        rec = C.create_string_buffer(VERIFY_RECORD_BYTES * max(n_local, 1))
        _check("lwkzg_verify_shard_begin",
               lib().lwkzg_verify_shard_begin(C.byref(self.h), rec, blobs, commitments_bytes, proofs_bytes, n_local, ts.ref()))
        self.records = rec.raw[:VERIFY_RECORD_BYTES * n_local]

    @classmethod
    def from_device(cls, blobs_ptr, comm_ptr, proofs_ptr, n_local, ts, stream=None):
        """the shard's inputs as device pointers (lwkzg_verify_shard_begin_device); the records come back to the host"""
        self = cls.__new__(cls)
        self.n = n_local
        self.ts = ts
        self.h = C.c_void_p()
        rec = C.create_string_buffer(VERIFY_RECORD_BYTES * max(n_local, 1))
        _check("lwkzg_verify_shard_begin_device",
               lib().lwkzg_verify_shard_begin_device(C.byref(self.h), rec, blobs_ptr, comm_ptr, proofs_ptr, n_local, ts.ref(), stream))
        self.records = rec.raw[:VERIFY_RECORD_BYTES * n_local]
        return self

    def partial(self, records_all, n_total, first_index):
        assert len(records_all) == VERIFY_RECORD_BYTES * n_total
        out = C.create_string_buffer(VERIFY_PARTIAL_BYTES)
        _check("lwkzg_verify_shard_partial", lib().lwkzg_verify_shard_partial(out, self.h, records_all, n_total, first_index))
        return out.raw

    def free(self):
        if self.h:
            lib().lwkzg_verify_shard_free(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def verify_shards_finish(partials, n_shards, n_total, ts):
    assert len(partials) == VERIFY_PARTIAL_BYTES * n_shards
    ok = C.c_bool(False)
    _check("lwkzg_verify_shards_finish", lib().lwkzg_verify_shards_finish(C.byref(ok), partials, n_shards, n_total, ts.ref()))
    return bool(ok.value)


# ---- batched host-pointer extensions --------------------------------------------------------------

def blob_to_kzg_commitment_batch(blobs, ts):
    n = len(blobs) // BYTES_PER_BLOB
    assert len(blobs) == n * BYTES_PER_BLOB
    out = C.create_string_buffer(48 * max(n, 1))
    bad = C.c_size_t(0)
    _check("lwkzg_blob_to_kzg_commitment_batch",
           lib().lwkzg_blob_to_kzg_commitment_batch(out, blobs, n, ts.ref(), C.byref(bad)))
    raw = out.raw      # (ONE copy of the buffer: `.raw` inside the comprehension copied all 48 n bytes per element -- 6 ms of Python per 4096 results)
    return [raw[48 * i:48 * i + 48] for i in range(n)]


def compute_blob_kzg_proof_batch(blobs, commitments, ts):
    n = len(blobs) // BYTES_PER_BLOB
    assert len(blobs) == n * BYTES_PER_BLOB and len(commitments) == 48 * n
    out = C.create_string_buffer(48 * max(n, 1))
    bad = C.c_size_t(0)
    _check("lwkzg_compute_blob_kzg_proof_batch",
           lib().lwkzg_compute_blob_kzg_proof_batch(out, blobs, commitments, n, ts.ref(), C.byref(bad)))
    raw = out.raw      # (ONE copy of the buffer: `.raw` inside the comprehension copied all 48 n bytes per element -- 6 ms of Python per 4096 results)
    return [raw[48 * i:48 * i + 48] for i in range(n)]


def compute_kzg_proof_batch(blobs, zs, ts):
    n = len(blobs) // BYTES_PER_BLOB
    assert len(blobs) == n * BYTES_PER_BLOB and len(zs) == 32 * n
    out, ys = C.create_string_buffer(48 * max(n, 1)), C.create_string_buffer(32 * max(n, 1))
    bad = C.c_size_t(0)
    _check("lwkzg_compute_kzg_proof_batch",
           lib().lwkzg_compute_kzg_proof_batch(out, ys, blobs, zs, n, ts.ref(), C.byref(bad)))
    raw, yraw = out.raw, ys.raw
    return [(raw[48 * i:48 * i + 48], yraw[32 * i:32 * i + 32]) for i in range(n)]


# ---- device-resident extensions (pointers are ints: torch tensor .data_ptr()) -----------------------

def blob_to_kzg_commitment_batch_device(out_ptr, blobs_ptr, n, ts, stream=None, status_ptr=None):
    _check("lwkzg_blob_to_kzg_commitment_batch_device",
           lib().lwkzg_blob_to_kzg_commitment_batch_device(out_ptr, blobs_ptr, n, ts.ref(), stream, status_ptr))


def compute_blob_kzg_proof_batch_device(out_ptr, blobs_ptr, comm_ptr, n, ts, stream=None, status_ptr=None):
    _check("lwkzg_compute_blob_kzg_proof_batch_device",
           lib().lwkzg_compute_blob_kzg_proof_batch_device(out_ptr, blobs_ptr, comm_ptr, n, ts.ref(), stream, status_ptr))


def commit_and_prove_batch_device(comm_ptr, proof_ptr, blobs_ptr, n, ts, stream=None, status_ptr=None):
    """Commitments and blob proofs of n device-resident blobs in one pass (the hash's commitment-independent part runs
    beside the commitment MSM); the same bytes as the two separate calls."""
    _check("lwkzg_commit_and_prove_batch_device",
           lib().lwkzg_commit_and_prove_batch_device(comm_ptr, proof_ptr, blobs_ptr, n, ts.ref(), stream, status_ptr))


def compute_challenges_device(z_ptr, blobs_ptr, comm_ptr, n, ts, stream=None):
    """z_i = compute_challenge(blob_i, commitment_i) for device-resident inputs (n x 32 bytes, mode's byte order)."""
    _check("lwkzg_compute_challenges_device",
           lib().lwkzg_compute_challenges_device(z_ptr, blobs_ptr, comm_ptr, n, ts.ref(), stream))


def g1_lincomb_setup_device(out_ptr, scalars_be_ptr, n_msm, ts, stream=None):
    _check("lwkzg_g1_lincomb_setup_device", lib().lwkzg_g1_lincomb_setup_device(out_ptr, scalars_be_ptr, n_msm, ts.ref(), stream))


def g1_msm_tiled_device(out_ptr, scalars_be_ptr, n_terms, ts, stream=None):
    _check("lwkzg_g1_msm_tiled_device", lib().lwkzg_g1_msm_tiled_device(out_ptr, scalars_be_ptr, n_terms, ts.ref(), stream))


def g1_sum_compressed(points48):
    n = len(points48) // 48
    out = C.create_string_buffer(48)
    _check("lwkzg_g1_sum_compressed", lib().lwkzg_g1_sum_compressed(out, points48, n))
    return out.raw


def fr_ntt4096_device(out_ptr, in_ptr, n, inverse, ts, stream=None):
    _check("lwkzg_fr_ntt4096_device", lib().lwkzg_fr_ntt4096_device(out_ptr, in_ptr, n, 1 if inverse else 0, ts.ref(), stream))


def pairing_product_is_one(g1_compressed, g2_compressed):
    """prod e(P_i, Q_i) == 1 for concatenated compressed points (host-only test hook)."""
    n = len(g1_compressed) // 48
    assert len(g1_compressed) == 48 * n and len(g2_compressed) == 96 * n
    ok = C.c_bool(False)
    _check("lwkzg_pairing_product_is_one", lib().lwkzg_pairing_product_is_one(C.byref(ok), g1_compressed, g2_compressed, n))
    return bool(ok.value)


def batch_challenge_host(records_all, n_total, mode):
    """r of verify_blob_kzg_proof_batch over a transcript of 160-byte records, canonical 32 bytes big-endian (lwkzg_batch_challenge_host)"""
    assert len(records_all) == VERIFY_RECORD_BYTES * n_total
    out = C.create_string_buffer(32)
    _check("lwkzg_batch_challenge_host", lib().lwkzg_batch_challenge_host(out, records_all, n_total, mode))
    return out.raw


def challenge_digests_host(blobs, commitments):
    n = len(commitments) // 48
    assert len(blobs) == n * BYTES_PER_BLOB
    out = C.create_string_buffer(32 * max(n, 1))
    _check("lwkzg_challenge_digests_host", lib().lwkzg_challenge_digests_host(out, blobs, commitments, n))
    raw = out.raw
    return [raw[32 * i:32 * i + 32] for i in range(n)]


def setup_image_bytes():
    return lib().lwkzg_setup_image_bytes()


def profile_enable(on=True):
    lib().lwkzg_profile_enable(1 if on else 0)


def profile_reset():
    lib().lwkzg_profile_reset()


def profile_report():
    n = lib().lwkzg_profile_report(None, 0)
    buf = C.create_string_buffer(n)
    lib().lwkzg_profile_report(buf, n)
    return json.loads(buf.value.decode())
