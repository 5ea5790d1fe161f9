"""Formula-generated blobs: the ten distinct blob byte strings of the c-kzg-4844 vectors
 plus the synthetic blobs bench.py and the parity tests use.

Data only -- no reference code. Element n of a c-kzg blob is serialised little-endian.
"""
import hashlib
import struct

R = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
N = 4096
BYTES_PER_BLOB = N * 32


def _gen(f, order="little"):
    return b"".join(int(f(n)).to_bytes(32, order) for n in range(N))


def _pow_blob(base):
    return _gen(lambda n: pow(base, n + 256, R))


_FORMULAS = {
    "zero": lambda: bytes(BYTES_PER_BLOB),
    "r_minus_1": lambda: _gen(lambda n: R - 1),
    "pow2": lambda: _pow_blob(2),
    "pow3": lambda: _pow_blob(3),
    "pow5": lambda: _pow_blob(5),
    "delta_3211": lambda: _gen(lambda n: 1 if n == 3211 else 0),
    "r_at_2111": lambda: _gen(lambda n: R if n == 2111 else 0),       # non-canonical element
    "all_ff": lambda: b"\xff" * BYTES_PER_BLOB,                        # non-canonical elements
    "pow2_plus_byte": lambda: _pow_blob(2) + b"\x00",                  # 131,073 bytes (wrong length)
    "pow2_minus_byte": lambda: _pow_blob(2)[:-1],                      # 131,071 bytes (wrong length)
}
BLOB_IDS = list(_FORMULAS)

# sha256 of each blob as it appears in the reference's YAML files (self-check)
BLOB_SHA256 = {
    "zero": "fa43239bcee7b97ca62f007cc68487560a39e19f74f3dde7486db3f98df8e471",
    "r_minus_1": "129f828bb4834048da378c0c244fd036426cfa473381712d3e5e85a31625d567",
    "pow2": "ca0aff662f7fa043cca957e6ad4daf7eb338264f5b99517d55703ff82c662040",
    "pow3": "a59392d15ecec9a8bf3f053099f7a5be819a60bf8554755fc438a08859135ad7",
    "pow5": "034378e3b29612a107944316014850ba52974bb10eda2e677a6462293e806e8e",
    "delta_3211": "efb94163c7bc9b5cc6f382a05d08349f8f61dcc31e604a20d6180aa8a914c512",
    "r_at_2111": "fcf52501475d2349f61ca179b09a983bbea86ef187f4d19b74ca33eff54c4edd",
    "all_ff": "b5a41c3758763bbec72769fab4a2533bf2db0b6312d93d25a695f9e4b9e02260",
    "pow2_plus_byte": "cc5336dfe1eb245720f9b9ac2e7a675f2bd7c40e13d5075b8c62ad3521dd60a8",
    "pow2_minus_byte": "ab3c6eca3f532b2dcdeb792c6a40e9188a0dc37b4e8c87f988366fb9e7b4fc78",
}

_cache = {}


def make_blob(name):
    """Bytes of a named c-kzg vector blob (little-endian elements)."""
    if name not in _cache:
        raw = _FORMULAS[name]()
        want = BLOB_SHA256.get(name)
        if want is not None:
            assert hashlib.sha256(raw).hexdigest() == want, name
        _cache[name] = raw
    return _cache[name]


def byteswap_elements(blob):
    """Reverse each 32-byte element (LE <-> BE view of the same integers)."""
    return b"".join(blob[i:i + 32][::-1] for i in range(0, len(blob), 32))


# ---------------------------------------------------------------- synthetic blobs

def _splitmix64(state):
    state = (state + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
    z = state
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    return state, z ^ (z >> 31)


def synthetic_blob(index, big_endian=True, full_range=False):
    """Blob `index` of the bench workload: every element is 31 bytes of a SplitMix64 stream
    seeded 0x4B5A47 + index, with one zero byte at the most-significant end of the 32 bytes
    (offset 0 big-endian / offset 31 little-endian) so every element is canonical (< 2^248 < r).
    Same construction as the reference's fuzz corpus generator (fuzz/gen_corpus/main.go:16-29)."""
    # full_range=True keeps all 32 random bytes: elements are then arbitrary 256-bit integers, which reference mode
    # reduces mod r (utils.rs:27-41) and c-kzg mode would reject -- the stress workload for the MSM's top window
    try:
        import numpy as np
        return _synthetic_blob_np(index, big_endian, np, full_range)
    except ImportError:
        pass
    st = (0x4B5A47 + index) & 0xFFFFFFFFFFFFFFFF
    words = []
    for _ in range(N * 4):
        st, v = _splitmix64(st)
        words.append(v)
    raw = struct.pack("<%dQ" % (N * 4), *words)
    out = bytearray(raw)
    if not full_range:
        for n in range(N):
            out[32 * n + (0 if big_endian else 31)] = 0
    return bytes(out)


def _synthetic_blob_np(index, big_endian, np, full_range=False):
    n = N * 4
    with np.errstate(over="ignore"):
        st = np.uint64((0x4B5A47 + index) & 0xFFFFFFFFFFFFFFFF) + np.uint64(0x9E3779B97F4A7C15) * np.arange(1, n + 1, dtype=np.uint64)
        z = st
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    out = z.astype("<u8").view(np.uint8).reshape(N, 32).copy()
    if not full_range:
        out[:, 0 if big_endian else 31] = 0
    return out.tobytes()


def synthetic_batch(first, count, big_endian=True, full_range=False):
    return b"".join(synthetic_blob(first + i, big_endian, full_range) for i in range(count))


def blob_scalars(blob, big_endian=True):
    order = "big" if big_endian else "little"
    return [int.from_bytes(blob[i:i + 32], order) for i in range(0, BYTES_PER_BLOB, 32)]
