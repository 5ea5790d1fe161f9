// fr_ops.hip -- scalar-field (Fr) side of the pipeline on gfx950.
//
//  * 4096-point radix-2 NTT/INTT, one workgroup per blob, whole transform in LDS (128 KiB of the
//    CU's 160 KiB). SURVEY a15: the reference never calls an FFT (KZGSettings.fs is always NULL,
//    /root/reference/src/lib.rs:754-758; the Lagrange conversion is commented out, :760-770); the
//    north star and the c-kzg-4844 semantics (evaluation-form blobs) need it.
//  * Polynomial::evaluate (Horner) + ruffini_division as ONE affine-map suffix scan per blob
//    (call sites /root/reference/src/lib.rs:320,329,389,394).
#include "kernels.h"

namespace lwk {

// omega = 7^((r-1)/4096) and its inverse, canonical little-endian limbs
__device__ __constant__ uint32_t kOmegaRaw[8] = {0xa5d36306u, 0xe206da11u, 0x378fbf96u, 0x0ad1347bu,
                                                 0xe0f8245fu, 0xfc3e8acfu, 0xa0f704f4u, 0x564c0a11u};
__device__ __constant__ uint32_t kOmegaInvRaw[8] = {0xd8543362u, 0x961a252du, 0x64183203u, 0x5046d178u,
                                                    0xb9dc5986u, 0x4ae25ffau, 0xc609b478u, 0x391b2856u};
// 4096^-1 mod r, canonical limbs (NOT Montgomery: multiplying a Montgomery value by it yields the raw product)
__device__ __constant__ uint32_t kNInvRaw[8] = {0x00100001u, 0x400fffffu, 0xbfce5c19u, 0xd3686828u,
                                                0x89213de7u, 0x5eb6a46au, 0xb46ae370u, 0x73e66878u};

// tw_fwd[k] = w^k, tw_inv[k] = w^-k, k < 2048 (Montgomery). One lane per entry: square-and-multiply.
__global__ __launch_bounds__(256) void k_build_twiddles(Fr *__restrict__ tw_fwd, Fr *__restrict__ tw_inv) {
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= kBlobElems / 2) return;
    uint32_t raw[8];
#pragma unroll
    for (int i = 0; i < 8; i++) raw[i] = kOmegaRaw[i];
    Fr w = fe_from_raw<FrParams>(raw);
#pragma unroll
    for (int i = 0; i < 8; i++) raw[i] = kOmegaInvRaw[i];
    Fr wi = fe_from_raw<FrParams>(raw);
    Fr a = Fr::one(), b = Fr::one();
    for (int bit = 10; bit >= 0; bit--) {
        a = sqr(a);
        b = sqr(b);
        if ((k >> bit) & 1) {
            a = a * w;
            b = b * wi;
        }
    }
    tw_fwd[k] = a;
    tw_inv[k] = b;
}

void launch_build_twiddles(Fr *tw_fwd, Fr *tw_inv, hipStream_t st) {
    ProfScope p("k_build_twiddles", st);
    hipLaunchKernelGGL(k_build_twiddles, dim3(kBlobElems / 2 / 256), dim3(256), 0, st, tw_fwd, tw_inv);
}

// ------------------------------------------------------------------------------------------------
// 4096-point DIT transform in LDS.  The input is consumed in the order given: feeding evaluations in
// bit-reversed order (exactly how a c-kzg-4844 blob stores them) yields natural-order output, so the
// usual bit-reversal pass disappears.
//
// Arithmetic: fr28.cuh (10 lazy limbs of 28 bits, Montgomery radix 2^280). An element enters through one product
// (x 2^256 -> x 2^280) and leaves through one (-> x 2^256, or -> x / 4096 in canonical words); in between a butterfly
// is ONE product, ten additions and ten offset subtractions, nothing is reduced or carried, except that the u inputs of
// stage 6 get one carry ripple: a subtraction adds two units of 2^28 to a limb, twelve stages of them would pass the
// fifteen a 32-bit limb holds (bounds: max 13 units per limb, value < 50 r, a product's column < 2^64; the same
// walk-through as a Python loop is in tests/test_capi_cpu.py).
// LDS: limbs 0..8 as nine arrays of 4096 words, limb 9 (< 2^9: the value's top) as 4096 halfwords: 152 KiB of the 160.

constexpr int kNttThreads = 1024;
constexpr int kNttLdsBytes = 9 * kBlobElems * 4 + kBlobElems * 2;

// tw28[k] = the Fr twiddle tw[k] in fr28 form (canonical limbs); one lane per entry
__global__ __launch_bounds__(256) void k_twiddles_to28(const Fr *__restrict__ tw, Fr28 *__restrict__ tw28, int n) {
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    tw28[k] = fr28_canonical(fr28_from_mont256(tw[k]));
}

void launch_twiddles_to28(const Fr *tw, Fr28 *tw28, hipStream_t st) {
    ProfScope p("k_twiddles_to28", st);
    hipLaunchKernelGGL(k_twiddles_to28, dim3(kBlobElems / 2 / 256), dim3(256), 0, st, tw, tw28, kBlobElems / 2);
}

struct NttLds {
    uint32_t *lo;   // [9][4096]
    uint16_t *top;  // [4096]
    __device__ Fr28 get(int i) const {
        Fr28 r;
#pragma unroll
        for (int j = 0; j < 9; j++) r.l[j] = lo[j * kBlobElems + i];
        r.l[9] = top[i];
        return r;
    }
    __device__ void put(int i, const Fr28 &v) const {
#pragma unroll
        for (int j = 0; j < 9; j++) lo[j * kBlobElems + i] = v.l[j];
        top[i] = (uint16_t)v.l[9];
    }
};

// FROM_BLOB = false: `in` holds Fr values in field.cuh's Montgomery form (the transform API); true: `in` is blob bytes,
// canonical little-endian evaluations in the order a c-kzg-4844 blob stores them (mode C): the parse, the range check
// (an element >= r marks its blob in `status`: c-kzg's C_KZG_BADARGS) and the entry into Montgomery form happen on load.
template <bool FROM_BLOB>
__global__ __launch_bounds__(kNttThreads) void k_ntt4096(const void *__restrict__ in, Fr *__restrict__ out,
                                                         const Fr28 *__restrict__ tw, int scale_to_raw,
                                                         int32_t *__restrict__ status) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    NttLds a{(uint32_t *)lds_raw, (uint16_t *)(lds_raw + 9 * kBlobElems * 4)};
    const int tid = threadIdx.x;
    const size_t base = (size_t)blockIdx.x * kBlobElems;
    auto load = [&](int i) {
        if constexpr (FROM_BLOB) {
            const uint4 *src = (const uint4 *)in + 2 * (base + i);
            uint4 lo = src[0], hi = src[1];
            uint32_t w[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
            if (raw_geq<8>(w, FrParams::MOD)) status[blockIdx.x] = kStatusBadArgs;  // benign race: same value
            return LWK_FR28_MUL_CONST(fr28_pack(w), RR);
        } else {
            return fr28_from_mont256(((const Fr *)in)[base + i]);
        }
    };
    auto store = [&](int i, const Fr28 &v) {
        if (scale_to_raw) fr28_to_raw_scaled(out[base + i].l, v);  // canonical words
        else out[base + i] = fr28_to_mont256(v);
    };
    for (int i = tid; i < kBlobElems; i += kNttThreads) a.put(i, load(i));
    __syncthreads();
    // stage s: butterflies of span half = 2^s; twiddle index = (k mod half) * (2048 / half). (Two stages per LDS round
    // trip -- radix-4 passes in registers, first and last pass straight from / to global memory -- measured the same
    // 0.38 ms per 1024 blobs: the kernel's time is its products.)
    for (int s = 0; s < 12; s++) {
        const int half = 1 << s;
        const int tshift = 11 - s;
        for (int bfly = tid; bfly < kBlobElems / 2; bfly += kNttThreads) {
            int k = bfly & (half - 1);
            int i0 = ((bfly >> s) << (s + 1)) + k;
            int i1 = i0 + half;
            Fr28 u = a.get(i0);
            if (s == 6) u = fr28_norm(u);  // the one carry ripple of the transform (see above)
            Fr28 v = a.get(i1);
            if (s != 0) v = fr28_mul(v, tw[k << tshift]);  // stage 0: every twiddle is 1, and v is a product's result already
            a.put(i0, fr28_add(u, v));
            a.put(i1, fr28_sub(u, v));
        }
        __syncthreads();
    }
    for (int i = tid; i < kBlobElems; i += kNttThreads) store(i, a.get(i));
}

template <bool FROM_BLOB>
static void launch_ntt_t(const void *in, Fr *out, const Fr28 *tw, int scale_to_raw, int32_t *status, size_t n_blobs, hipStream_t st) {
    ProfScope p("k_ntt4096", st);
    static const hipError_t attr_set =
        hipFuncSetAttribute((const void *)k_ntt4096<FROM_BLOB>, hipFuncAttributeMaxDynamicSharedMemorySize, kNttLdsBytes);
    (void)attr_set;
    hipLaunchKernelGGL(k_ntt4096<FROM_BLOB>, dim3((unsigned)n_blobs), dim3(kNttThreads), kNttLdsBytes, st, in, out, tw, scale_to_raw,
                       status);
}

void launch_ntt4096(const Fr *in, Fr *out, const Fr28 *tw, int inverse_scale_to_raw, size_t n_blobs, hipStream_t st) {
    launch_ntt_t<false>(in, out, tw, inverse_scale_to_raw, nullptr, n_blobs, st);
}

// mode C front end in one launch: blob bytes (canonical little-endian evaluations, bit-reversed domain order) ->
// canonical monomial coefficients; status[b] = kStatusBadArgs for a blob with an element >= r
void launch_blob_evaluations_to_coefficients(const uint8_t *blobs, uint32_t *coeffs_raw, const Fr28 *tw_inv, int32_t *status,
                                             size_t n_blobs, hipStream_t st) {
    launch_ntt_t<true>(blobs, (Fr *)coeffs_raw, tw_inv, 1, status, n_blobs, st);
}

__global__ __launch_bounds__(256) void k_bitrev_permute(const Fr *__restrict__ in, Fr *__restrict__ out, size_t n) {
    size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n) return;
    uint32_t i = (uint32_t)(g & (kBlobElems - 1));
    uint32_t r = __brev(i) >> 20;
    out[(g - i) + r] = in[g];
}

void launch_bitrev_permute(const Fr *in, Fr *out, size_t n_blobs, hipStream_t st) {
    ProfScope p("k_bitrev_permute", st);
    size_t n = n_blobs * kBlobElems;
    hipLaunchKernelGGL(k_bitrev_permute, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, in, out, n);
}

__global__ __launch_bounds__(256) void k_fr_be_to_mont(const uint4 *__restrict__ in, Fr *__restrict__ out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint4 hi = in[2 * i], lo = in[2 * i + 1];
    uint32_t s[8] = {__builtin_bswap32(lo.w), __builtin_bswap32(lo.z), __builtin_bswap32(lo.y), __builtin_bswap32(lo.x),
                     __builtin_bswap32(hi.w), __builtin_bswap32(hi.z), __builtin_bswap32(hi.y), __builtin_bswap32(hi.x)};
    out[i] = fe_from_raw<FrParams>(s);
}
void launch_fr_be_to_mont(const uint8_t *in_be, Fr *out, size_t n_elems, hipStream_t st) {
    ProfScope p("k_fr_be_to_mont", st);
    hipLaunchKernelGGL(k_fr_be_to_mont, dim3((unsigned)((n_elems + 255) / 256)), dim3(256), 0, st, (const uint4 *)in_be,
                       out, n_elems);
}

__global__ __launch_bounds__(256) void k_fr_mont_to_be(const Fr *__restrict__ in, uint4 *__restrict__ out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t s[8];
    fe_to_raw<FrParams>(s, in[i]);
    out[2 * i] = make_uint4(__builtin_bswap32(s[7]), __builtin_bswap32(s[6]), __builtin_bswap32(s[5]),
                            __builtin_bswap32(s[4]));
    out[2 * i + 1] = make_uint4(__builtin_bswap32(s[3]), __builtin_bswap32(s[2]), __builtin_bswap32(s[1]),
                                __builtin_bswap32(s[0]));
}
void launch_fr_mont_to_be(const Fr *in, uint8_t *out_be, size_t n_elems, hipStream_t st) {
    ProfScope p("k_fr_mont_to_be", st);
    hipLaunchKernelGGL(k_fr_mont_to_be, dim3((unsigned)((n_elems + 255) / 256)), dim3(256), 0, st, in, (uint4 *)out_be,
                       n_elems);
}

__global__ __launch_bounds__(256) void k_raw_to_be(const uint4 *__restrict__ in, uint4 *__restrict__ out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint4 lo = in[2 * i], hi = in[2 * i + 1];
    out[2 * i] = make_uint4(__builtin_bswap32(hi.w), __builtin_bswap32(hi.z), __builtin_bswap32(hi.y),
                            __builtin_bswap32(hi.x));
    out[2 * i + 1] = make_uint4(__builtin_bswap32(lo.w), __builtin_bswap32(lo.z), __builtin_bswap32(lo.y),
                                __builtin_bswap32(lo.x));
}
void launch_raw_to_be(const uint32_t *raw, uint8_t *out_be, size_t n_elems, hipStream_t st) {
    ProfScope p("k_raw_to_be", st);
    hipLaunchKernelGGL(k_raw_to_be, dim3((unsigned)((n_elems + 255) / 256)), dim3(256), 0, st, (const uint4 *)raw,
                       (uint4 *)out_be, n_elems);
}

// ------------------------------------------------------------------------------------------------
// y = p(z), q = (p - y) / (x - z)
//
// Horner: acc_i = c_i + z * acc_{i+1}, acc_4096 = 0.  Then y = acc_0 and q_{i-1} = acc_i (Ruffini).
// The recurrence is a composition of affine maps x -> c + m x, so it parallelises as a suffix scan:
// lane t owns coefficients [16t, 16t+16); L_t = sum_k c_{16t+k} z^k; H_t = L_t + z^16 H_{t+1}
// (Hillis-Steele over (multiplier, value) pairs in LDS); acc at the chunk's upper edge is H_{t+1}.
//
// Representation: only z and its powers (the multipliers) are in Montgomery form. The coefficients, every accumulator
// and the quotient stay RAW canonical integers: a Montgomery product of a Montgomery multiplier (z R) with a raw value
// x is (z R) x / R = z x, raw again -- so the 4096 conversions in and the 4096 conversions out (a product each) that a
// uniform Montgomery pipeline would need never happen. The coefficients arrive reduced (k_parse_be_reduce, the
// inverse transform's exit), which is what the product's "first operand < r" requirement and the additions need.

// Arithmetic (r05): fr28.cuh -- 10 lazy limbs of 28 bits, a product is 200 multiply-adds with no carry handling (the 8 x 32-bit CIOS
// form this kernel used needs ~2x the instructions); sums keep their carries; a quotient coefficient (< 3r: c + z acc with the product
// < 2r) is brought to its canonical form by one carry ripple and two conditional subtractions. Bounds: every product has the
// multiplier (z or a power of z: a product's result, normalised, < 2r) on one side and a value of at most 24 limb units / 2^24 r
// on the other; the scan's values grow by one product result per level (<= 3r + 10 x 2r, limbs <= 12 units).
//
// 256 lanes x 16 coefficients: 51 products deep.

// value < 4r, lazy limbs -> canonical (< r): one ripple, then r is taken off at most three times
__device__ __forceinline__ Fr28 fr28_canonical_lazy(const Fr28 &a) {
    Fr28 v = fr28_norm(a);
#pragma unroll
    for (int rep = 0; rep < 3; rep++) {
        Fr28 d;
        uint32_t borrow = 0;
#pragma unroll
        for (int i = 0; i < 10; i++) {
            const uint32_t t = v.l[i] - R28::MOD[i] - borrow;
            borrow = t >> 31;
            d.l[i] = t & R28::MASK;
        }
#pragma unroll
        for (int i = 0; i < 10; i++) v.l[i] = borrow ? v.l[i] : d.l[i];
    }
    return v;
}

template <int kThreads>
__global__ __launch_bounds__(kThreads) void k_eval_quotient(const uint4 *__restrict__ coeffs_raw, const Fr *__restrict__ z_mont,
                                                            uint4 *__restrict__ quot_raw, uint8_t *__restrict__ y_out, int le,
                                                            const uint32_t *__restrict__ only_if, int from_blob_be) {
    constexpr int kChunk = kBlobElems / kThreads;
    __shared__ Fr28 sh_m[kThreads];
    __shared__ Fr28 sh_v[kThreads];
    const int t = threadIdx.x;
    const size_t blob = blockIdx.x;
    if (only_if && !only_if[blob]) return;   // a second pass over the blobs whose challenge changed (engine.hip: blob_proof_batch_device)
    const uint4 *cin = coeffs_raw + (blob * kBlobElems + (size_t)t * kChunk) * 2;
    const Fr28 z = fr28_from_mont256(z_mont[blob]);   // z 2^280, normalised, < 2r

    Fr28 c[kChunk];
#pragma unroll
    for (int k = 0; k < kChunk; k++) {
        const uint4 lo = cin[2 * k], hi = cin[2 * k + 1];
        if (from_blob_be) {   // `coeffs_raw` is the reference-mode BLOB itself: big-endian elements, reduced mod r as k_parse_be_reduce does (2^256 < 3r)
            uint32_t sw[8] = {__builtin_bswap32(hi.w), __builtin_bswap32(hi.z), __builtin_bswap32(hi.y), __builtin_bswap32(hi.x),
                              __builtin_bswap32(lo.w), __builtin_bswap32(lo.z), __builtin_bswap32(lo.y), __builtin_bswap32(lo.x)};
#pragma unroll
            for (int rep = 0; rep < 2; rep++) {
                uint32_t d[8];
                const uint32_t br = raw_sub<8>(d, sw, FrParams::MOD);
#pragma unroll
                for (int j = 0; j < 8; j++) sw[j] = br ? sw[j] : d[j];
            }
            c[k] = fr28_pack(sw);
        } else {
            const uint32_t w[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
            c[k] = fr28_pack(w);  // raw, canonical
        }
    }
    Fr28 L = c[kChunk - 1];
#pragma unroll
    for (int k = kChunk - 2; k >= 0; k--) L = fr28_add(fr28_mul(z, L), c[k]);   // < 3r, limbs < 2 units
    Fr28 m = z;
#pragma unroll
    for (int k = 1; k < kChunk; k <<= 1) m = fr28_mul(m, m);  // z^kChunk

    Fr28 v = L;
    if (!quot_raw) {
        // y = p(z) alone (batch verification: r06). No suffix is wanted, so the scan's 8 levels x 2 products on every lane become a TREE on
        // compacted lanes: lane t < active takes elements 2t and 2t + 1, v' = v_2t + m v_2t+1 with m = z^(16 2^level) -- the same for every
        // pair of a level --, and squares its m. From the second level on only wave 0 works (128 -> 64 -> ... -> 1 active lanes): 18
        // wave-products behind the 19 of the chunk where the scan issued 64. Bounds as in the scan: a value grows by one product result per
        // level (<= 3r + 8 x 2r, limbs <= 10 units) and enters the next level's product as the lazy factor.
        sh_v[t] = v;
        __syncthreads();
        for (int active = kThreads / 2; active >= 1; active >>= 1) {
            Fr28 a, b;
            if (t < active) {
                a = sh_v[2 * t];
                b = sh_v[2 * t + 1];
            }
            __syncthreads();
            if (t < active) {
                v = fr28_add(a, fr28_mul(m, b));
                sh_v[t] = v;
                if (active > 1) m = fr28_mul(m, m);
            }
            __syncthreads();
        }
        if (t == 0 && y_out) {
            const Fr28 yv = fr28_canonical(LWK_FR28_MUL_CONST(v, ONE));
            uint32_t w[8];
            fr28_unpack(w, yv);
            uint8_t *yo = y_out + 32 * blob;
            if (le) raw_to_le<8>(yo, w); else raw_to_be<8>(yo, w);
        }
        return;
    }
    sh_m[t] = m;
    sh_v[t] = v;
    __syncthreads();
    for (int d = 1; d < kThreads; d <<= 1) {
        const bool has = t + d < kThreads;
        Fr28 om, ov;
        if (has) {
            om = sh_m[t + d];
            ov = sh_v[t + d];
        }
        __syncthreads();
        if (has) {
            v = fr28_add(v, fr28_mul(m, ov));   // (one more product result per level: the value and limb bounds above)
            m = fr28_mul(m, om);
        }
        sh_m[t] = m;
        sh_v[t] = v;
        __syncthreads();
    }
    // v == H_t
    Fr28 acc;
    if (t + 1 < kThreads) {
        acc = fr28_canonical(LWK_FR28_MUL_CONST(sh_v[t + 1], ONE));   // (up to 3r + 10 x 2r with lazy limbs: one product by 2^280 mod r brings it under 2r)
    } else {
#pragma unroll
        for (int i = 0; i < 10; i++) acc.l[i] = 0;
    }
    uint4 *qout = quot_raw + (blob * kBlobElems) * 2;
    const int i0 = t * kChunk;
#pragma unroll
    for (int k = kChunk - 1; k >= 0; k--) {
        acc = fr28_canonical_lazy(fr28_add(c[k], fr28_mul(z, acc)));
        const int i = i0 + k;
        if (i >= 1) {
            uint32_t w[8];
            fr28_unpack(w, acc);
            qout[2 * (i - 1)] = make_uint4(w[0], w[1], w[2], w[3]);
            qout[2 * (i - 1) + 1] = make_uint4(w[4], w[5], w[6], w[7]);
        }
    }
    if (t == kThreads - 1) {
        qout[2 * (kBlobElems - 1)] = make_uint4(0, 0, 0, 0);
        qout[2 * (kBlobElems - 1) + 1] = make_uint4(0, 0, 0, 0);
    }
    if (t == 0 && y_out) {
        uint32_t w[8];
        fr28_unpack(w, acc);  // acc_0 = y, raw and canonical
        uint8_t *yo = y_out + 32 * blob;
        if (le) raw_to_le<8>(yo, w); else raw_to_be<8>(yo, w);
    }
}

void launch_eval_quotient(const uint32_t *coeffs_raw, const Fr *z_mont, uint32_t *quot_raw, uint8_t *y_out, int le,
                          size_t n_blobs, hipStream_t st, const uint32_t *only_if) {
    ProfScope p(only_if ? "k_eval_quotient_redo" : "k_eval_quotient", st);
    // (512 and 1024 lanes per blob -- a chain of 37 / 29 products instead of 51 -- measured SLOWER for one blob, 0.061 / 0.077 ms against 0.054:
    // the barriers of 8 / 16 waves cost more than the shorter chain saves; gpurun_out r05)
    hipLaunchKernelGGL(k_eval_quotient<256>, dim3((unsigned)n_blobs), dim3(256), 0, st, (const uint4 *)coeffs_raw, z_mont, (uint4 *)quot_raw,
                       y_out, le, only_if, 0);
}

// y = p(z) of n reference-mode blobs straight from their bytes (big-endian coefficients), no workspace: the evaluation of a
// device-resident batch verification as ONE launch (r05; it was a parse and an evaluation per chunk of 1024)
void launch_eval_y_from_blobs_be(const uint8_t *blobs, const Fr *z_mont, uint8_t *y_out, size_t n_blobs, hipStream_t st) {
    ProfScope p("k_eval_quotient_from_blobs", st);
    hipLaunchKernelGGL(k_eval_quotient<256>, dim3((unsigned)n_blobs), dim3(256), 0, st, (const uint4 *)blobs, z_mont, (uint4 *)nullptr, y_out, 0,
                       (const uint32_t *)nullptr, 1);
}

// ---- the quotient in EVALUATION form (r05) --------------------------------------------------------------------
//
// c-kzg mode on the Lagrange form of the setup: the blob IS the polynomial's evaluations p_i = p(w_i) on the bit-reversed domain
// (w_i = w^bitrev12(i)), and the proof is the MSM of the QUOTIENT's evaluations q_i = (p_i - y) / (w_i - z) over [l_i(tau)]G -- no
// transform anywhere. With inv_i = 1 / (z - w_i) (ONE inversion per blob: Montgomery's trick as a product tree over the workgroup,
// its root inverted by division steps on one lane),
//     y   = (z^4096 - 1) / 4096 * sum_i p_i w_i inv_i          (barycentric formula; w_i inv_i = z inv_i - 1, so the sum is
//                                                               z sum_i p_i inv_i - sum_i p_i: no product by w_i)
//     q_i = (y - p_i) inv_i
// and when z IS a domain point w_m (the root of the tree is zero: found, d_m replaced by 1, the tree built again):
//     y = p_m,   q_m = -(1 / w_m) sum_(i != m) q_i w_i         (the limit of the same quotient; c-kzg-4844's compute_kzg_proof_impl)
// Arithmetic on fr28.cuh: z, w_i, inv_i in Montgomery form (x 2^280), p_i, y, q_i plain; bounds in the comments as (value in units of r,
// limb in units of 2^28). 5 products per element + 2 per tree node, against 3 (Horner + Ruffini) + 2 x 7.5 (two transforms).
__device__ __forceinline__ Fr28 fr28_neg_canonical(const Fr28 &a) {   // a canonical -> (r - a) mod r, canonical
    Fr28 d;
    uint32_t borrow = 0, nz = 0;
#pragma unroll
    for (int i = 0; i < 10; i++) {
        const uint32_t t = R28::MOD[i] - a.l[i] - borrow;
        borrow = t >> 31;
        d.l[i] = t & R28::MASK;
        nz |= a.l[i];
    }
#pragma unroll
    for (int i = 0; i < 10; i++) d.l[i] = nz ? d.l[i] : 0u;
    return d;
}
__device__ __forceinline__ Fr28 fr28_const_one() {
    Fr28 r;
#pragma unroll
    for (int i = 0; i < 10; i++) r.l[i] = R28::ONE[i];
    return r;
}
// roots[i] = w_i = w^bitrev12(i), canonical Montgomery form, for all 4096 i IN ELEMENT ORDER (k_roots_brp28 below writes the table once
// per setup, behind the transform's 2048 twiddles): thread t of a workgroup works on the elements i = 256 k + t, so that the 64 lanes
// of a wave read 64 CONSECUTIVE table entries, blob elements and quotient slots (2.5 KB / 2 KB per access instead of 64 cache lines)
__global__ __launch_bounds__(256) void k_roots_brp28(const Fr28 *__restrict__ tw28, Fr28 *__restrict__ roots) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= kBlobElems) return;
    const uint32_t e = __brev(i) >> 20;
    const Fr28 w = tw28[e & (kBlobElems / 2 - 1)];   // tw[j] = w^j for j < 2048, w^(j + 2048) = -w^j
    Fr28 r = w;
    if (e >= kBlobElems / 2) {
        uint32_t borrow = 0;
#pragma unroll
        for (int j = 0; j < 10; j++) {
            const uint32_t t = R28::MOD[j] - w.l[j] - borrow;
            borrow = t >> 31;
            r.l[j] = t & R28::MASK;
        }
    }
    roots[i] = r;
}
void launch_roots_brp28(const Fr28 *tw28, Fr28 *roots, hipStream_t st) {
    hipLaunchKernelGGL(k_roots_brp28, dim3(kBlobElems / 256), dim3(256), 0, st, tw28, roots);
}
__device__ __forceinline__ Fr28 evalform_load(const uint4 *__restrict__ p) {
    const uint4 lo = p[0], hi = p[1];
    const uint32_t w[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    return fr28_pack(w);
}
// block sum of one lazy value per thread (normalised limbs in, value bound v): the sum of 256 in `out` of every thread, bound 256 v
template <int kThreads>
__device__ __forceinline__ Fr28 evalform_block_sum(Fr28 *sh, const Fr28 &mine, int t) {
    __syncthreads();   // (sh is the tree's storage: every thread has read what it needed)
    sh[t] = mine;
    __syncthreads();
    for (int d = kThreads / 2; d >= 1; d >>= 1) {
        if (t < d) sh[t] = fr28_norm(fr28_add(sh[t], sh[t + d]));
        __syncthreads();
    }
    return sh[0];
}

// two sums at once (sh holds 2 x kThreads values)
template <int kThreads>
__device__ __forceinline__ void evalform_block_sum2(Fr28 *sh, Fr28 &a, Fr28 &b, int t) {
    __syncthreads();
    sh[t] = a;
    sh[kThreads + t] = b;
    __syncthreads();
    for (int d = kThreads / 2; d >= 1; d >>= 1) {
        if (t < d) {
            sh[t] = fr28_norm(fr28_add(sh[t], sh[t + d]));
            sh[kThreads + t] = fr28_norm(fr28_add(sh[kThreads + t], sh[kThreads + t + d]));
        }
        __syncthreads();
    }
    a = sh[0];
    b = sh[kThreads];
}

template <int kThreads>
__global__ __launch_bounds__(kThreads) void k_eval_quotient_evalform(const uint4 *__restrict__ evals_raw, const Fr *__restrict__ z_mont,
                                                                     const Fr28 *__restrict__ roots, uint4 *__restrict__ quot_raw,
                                                                     uint8_t *__restrict__ y_out, int le, const uint32_t *__restrict__ only_if,
                                                                     int32_t *__restrict__ status) {
    constexpr int kChunk = kBlobElems / kThreads;
    static_assert(kChunk == 16 && kThreads == 256, "the index arithmetic below is for 256 x 16");
    __shared__ Fr28 tree[2 * kThreads];   // heap order: node j has children 2j, 2j + 1; leaf of thread t = tree[kThreads + t]
    __shared__ Fr28 sh_c;
    __shared__ int sh_m;
    __shared__ uint32_t sh_zero;
    const int t = threadIdx.x;
    const size_t blob = blockIdx.x;
    if (only_if && !only_if[blob]) return;
    // element k of this thread is i = 256 k + t (see k_roots_brp28)
    const uint4 *pin = evals_raw + (blob * kBlobElems + (size_t)t) * 2;
    if (status) {   // `evals_raw` is the c-kzg BLOB itself (its canonical little-endian elements are the evaluations): the front end's range check
                    // (k_copy_le_check's, when the blob goes through the workspace) is made here -- an element >= r: BADARGS for the blob
        bool bad = false;
#pragma unroll 4
        for (int k = 0; k < kChunk; k++) {
            const uint4 lo = pin[2 * k * kThreads], hi = pin[2 * k * kThreads + 1];
            const uint32_t w[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
            bad = bad || raw_geq<8>(w, FrParams::MOD);
        }
        if (bad) status[blob] = kStatusBadArgs;   // (benign race: same value)
    }
    const Fr28 z = fr28_from_mont256(z_mont[blob]);   // (2, 1)
    const Fr28 one = fr28_const_one();
    if (t == 0) sh_m = -1;
    // (z^4096 - 1) / 4096, Montgomery form, (2, 1): by ONE lane, of a wave that does not invert below. Left to every thread the compiler
    // computes these thirteen products of workgroup-uniform values on the SCALAR unit, 19.5k scalar instructions per wave (rocprofv3 PMC,
    // gpurun_out r05/pmc_evf: as many as two thirds of the kernel's vector instructions)
    if (t == (int)(((blockIdx.x + 2u) & 3u) * 64u)) {
        Fr28 zn = z;
#pragma unroll
        for (int j = 0; j < 10; j++) asm volatile("" : "+v"(zn.l[j]));   // (opaque, in vector registers: the chain below stays on the vector unit)
#pragma unroll 1
        for (int k = 0; k < 12; k++) zn = fr28_mul(zn, zn);
        sh_c = LWK_FR28_MUL_CONST(fr28_sub(zn, one), NINV_M);
    }
    __syncthreads();

    // (the three loops over the 16 elements are NOT unrolled: 100 products of 230 instructions would be three times the instruction cache;
    // pre[] lives in scratch, 40 bytes in and out per product)
    Fr28 pre[kChunk];   // first the prefix products d_0 .. d_k, then (in place, from the top) inv_k
    // prefix products, the product tree and its root's inverse; `m_`: the index whose d is replaced by 1 (none: -1). True: the root was zero.
    auto build = [&](const int m_) -> bool {
#pragma unroll 1
        for (int k = 0; k < kChunk; k++) {   // d_k = z - w_k: (6, 3)
            const Fr28 d = k * kThreads + t == m_ ? one : fr28_sub(z, roots[k * kThreads + t]);
            pre[k] = k == 0 ? d : fr28_mul(pre[k - 1], d);   // (2, 1) x (6, 3)
        }
        tree[kThreads + t] = pre[kChunk - 1];
        __syncthreads();
#pragma unroll 1
        for (int w = kThreads / 2; w >= 1; w >>= 1) {   // nodes w .. 2w - 1
            if (t < w) tree[w + t] = fr28_mul(tree[2 * (w + t)], tree[2 * (w + t) + 1]);
            __syncthreads();
        }
        // the root's inverse, by one lane (of a different wave from workgroup to workgroup: the four workgroups of a compute unit
        // then invert on four different SIMDs)
        if (t == (int)((blockIdx.x & 3u) * 64u)) {
            const Fr28 rc = fr28_canonical(tree[1]);
            uint32_t x[8], xi[8], nz = 0;
            fr28_unpack(x, rc);
#pragma unroll
            for (int k = 0; k < 8; k++) nz |= x[k];
            fr_inv_raw32(xi, x);
            tree[1] = LWK_FR28_MUL_CONST(fr28_pack(xi), R3);   // 1 / (P 2^280) -> 2^280 / P
            sh_zero = nz == 0;
        }
        __syncthreads();
        return sh_zero != 0;
    };
    int m = -1;         // the index whose w_m == z, if any (uniform over the workgroup)
    if (build(-1)) {    // z is a domain point: which one?
        const Fr28 zc = fr28_canonical(z);
#pragma unroll 1
        for (int k = 0; k < kChunk; k++) {
            const Fr28 w = roots[k * kThreads + t];
            uint32_t diff = 0;
#pragma unroll
            for (int j = 0; j < 10; j++) diff |= w.l[j] ^ zc.l[j];
            if (!diff) sh_m = k * kThreads + t;
        }
        __syncthreads();
        m = sh_m;
        __syncthreads();
        (void)build(m);
    }
    // down: the inverse of a node = the parent's inverse times the sibling
    for (int w = 1; w < kThreads; w <<= 1) {   // parents w .. 2w - 1
        if (t < w) {
            const Fr28 ip = tree[w + t], l = tree[2 * (w + t)], r = tree[2 * (w + t) + 1];
            tree[2 * (w + t)] = fr28_mul(ip, r);
            tree[2 * (w + t) + 1] = fr28_mul(ip, l);
        }
        __syncthreads();
    }
    // inv_k in place of the prefix products, from the top; the barycentric sum on the way
    Fr28 run = tree[kThreads + t];   // 1 / (d_0 .. d_15)
    Fr28 acc, psum;   // sum of p_k inv_k, sum of p_k (both plain)
#pragma unroll
    for (int j = 0; j < 10; j++) acc.l[j] = psum.l[j] = 0;
#pragma unroll 1
    for (int k = kChunk - 1; k >= 0; k--) {
        Fr28 inv_k = run;
        if (k > 0) {
            inv_k = fr28_mul(run, pre[k - 1]);
            const Fr28 d = k * kThreads + t == m ? one : fr28_sub(z, roots[k * kThreads + t]);
            run = fr28_mul(run, d);
        }
        pre[k] = inv_k;
        if (m < 0) {   // (uniform; on the domain y is p_m and the sums are not needed)
            const Fr28 pk = evalform_load(pin + 2 * k * kThreads);
            acc = fr28_add(acc, fr28_mul(pk, inv_k));   // (2, 1) each
            psum = fr28_add(psum, pk);                  // (1, 1) each
            if (k == kChunk / 2) {                      // (8 x 2 units of limb at most between ripples)
                acc = fr28_norm(acc);
                psum = fr28_norm(psum);
            }
        }
    }
    Fr28 yc;
    if (m < 0) {
        acc = fr28_norm(acc);     // (32, 1)
        psum = fr28_norm(psum);   // (16, 1)
        evalform_block_sum2<kThreads>(tree, acc, psum, t);   // (8192, 1), (4096, 1)
        const Fr28 a = fr28_sub(fr28_mul(acc, z), LWK_FR28_MUL_CONST(psum, ONE));   // z sum p inv - sum p: (6, 3)
        yc = fr28_canonical(fr28_mul(a, sh_c));
    } else {
        __syncthreads();
        if ((m & (kThreads - 1)) == t) tree[0] = evalform_load(pin + 2 * (m - t));
        __syncthreads();
        yc = tree[0];   // p_m
    }
    if (!quot_raw) {   // only y = p(z) is wanted (batch verification)
        if (t == 0 && y_out) {
            uint32_t wd[8];
            fr28_unpack(wd, yc);
            uint8_t *yo = y_out + 32 * blob;
            if (le) raw_to_le<8>(yo, wd); else raw_to_be<8>(yo, wd);
        }
        return;
    }
    uint4 *qout = quot_raw + (blob * kBlobElems + (size_t)t) * 2;
    Fr28 part;   // z on the domain: sum of q_i w_i over this thread's i != m
#pragma unroll
    for (int j = 0; j < 10; j++) part.l[j] = 0;
#pragma unroll 1
    for (int k = 0; k < kChunk; k++) {
        const Fr28 q = fr28_canonical(fr28_mul(fr28_sub(yc, evalform_load(pin + 2 * k * kThreads)), pre[k]));   // (6, 3) x (2, 1)
        if (k * kThreads + t != m) {
            uint32_t wd[8];
            fr28_unpack(wd, q);
            qout[2 * k * kThreads] = make_uint4(wd[0], wd[1], wd[2], wd[3]);
            qout[2 * k * kThreads + 1] = make_uint4(wd[4], wd[5], wd[6], wd[7]);
            if (m >= 0) {
                part = fr28_add(part, fr28_mul(q, roots[k * kThreads + t]));
                if (k == kChunk / 2) part = fr28_norm(part);
            }
        }
    }
    if (m >= 0) {   // (uniform)
        const Fr28 sum = evalform_block_sum<kThreads>(tree, fr28_norm(part), t);
        if ((m & (kThreads - 1)) == t) {
            const uint32_t e = __brev((uint32_t)m) >> 20, j = (kBlobElems - e) & (kBlobElems - 1);   // 1 / w^e = w^(4096 - e): the element whose exponent that is
            const Fr28 winv = roots[__brev(j) >> 20];
            const Fr28 qm = fr28_neg_canonical(fr28_canonical(fr28_mul(sum, winv)));
            uint32_t wd[8];
            fr28_unpack(wd, qm);
            qout[2 * (m - t)] = make_uint4(wd[0], wd[1], wd[2], wd[3]);
            qout[2 * (m - t) + 1] = make_uint4(wd[4], wd[5], wd[6], wd[7]);
        }
    }
    if (t == 0 && y_out) {
        uint32_t wd[8];
        fr28_unpack(wd, yc);
        uint8_t *yo = y_out + 32 * blob;
        if (le) raw_to_le<8>(yo, wd); else raw_to_be<8>(yo, wd);
    }
}

void launch_eval_quotient_evalform(const uint32_t *evals_raw, const Fr *z_mont, const Fr28 *roots_brp28, uint32_t *quot_raw, uint8_t *y_out, int le,
                                   size_t n_blobs, hipStream_t st, const uint32_t *only_if) {
    ProfScope p(only_if ? "k_eval_quotient_evalform_redo" : "k_eval_quotient_evalform", st);
    hipLaunchKernelGGL(k_eval_quotient_evalform<256>, dim3((unsigned)n_blobs), dim3(256), 0, st, (const uint4 *)evals_raw, z_mont, roots_brp28,
                       (uint4 *)quot_raw, y_out, le, only_if, (int32_t *)nullptr);
}

// y = p(z) of n c-kzg-mode blobs straight from their bytes (little-endian evaluations; an element >= r: status[blob] = BADARGS), no workspace,
// any number in one launch: the evaluation of a device-resident batch verification on the Lagrange form
void launch_eval_y_from_blobs_evalform(const uint8_t *blobs, const Fr *z_mont, const Fr28 *roots_brp28, uint8_t *y_out, int32_t *status,
                                       size_t n_blobs, hipStream_t st) {
    ProfScope p("k_eval_quotient_evalform_from_blobs", st);
    hipLaunchKernelGGL(k_eval_quotient_evalform<256>, dim3((unsigned)n_blobs), dim3(256), 0, st, (const uint4 *)blobs, z_mont, roots_brp28,
                       (uint4 *)nullptr, y_out, 1, (const uint32_t *)nullptr, status);
}

// flags[i] = the 48 bytes at a + 48 i differ from those at b + 48 i (a commitment whose canonical encoding is not what the caller sent)
__global__ __launch_bounds__(256) void k_flag_differs48(const uint32_t *__restrict__ a, const uint32_t *__restrict__ b,
                                                        uint32_t *__restrict__ flags, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t d = 0;
#pragma unroll
    for (int k = 0; k < 12; k++) d |= a[12 * i + k] ^ b[12 * i + k];
    flags[i] = d != 0;
}
void launch_flag_differs48(const uint8_t *a, const uint8_t *b, uint32_t *flags, size_t n, hipStream_t st) {
    ProfScope p("k_flag_differs48", st);
    hipLaunchKernelGGL(k_flag_differs48, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const uint32_t *)a, (const uint32_t *)b, flags, n);
}

// ---- c-kzg mode without the transform: the Lagrange form of the setup ------------------------------------------
//
// L_i = [l_i(tau)]G, l_i the Lagrange polynomial of the domain point w_i = w^bitrev12(i), is the commitment of l_i's MONOMIAL
// coefficients c_(i,k) = w_i^(-k) / 4096 -- a column of the inverse DFT matrix -- over the monomial setup, so the 4096 points are 4096
// commitments of the existing engine (engine.hip: lagrange_prepare). This kernel writes those 4096 "blobs" of coefficients, in the
// canonical raw form the MSM's digit extraction reads: row b of the output = l_(first + b). One workgroup per row, 16 consecutive
// powers per thread (a 12-step square-and-multiply to the thread's first power, then 15 products).
__global__ __launch_bounds__(256) void k_idft_columns(uint4 *__restrict__ coeffs_raw, const Fr *__restrict__ tw_inv, uint32_t first, Fr ninv) {
    const uint32_t i = first + blockIdx.x, t = threadIdx.x;
    const uint32_t e = __brev(i) >> 20;                                   // w_i = w^e
    const Fr base = e < kBlobElems / 2 ? tw_inv[e] : neg(tw_inv[e - kBlobElems / 2]);   // w^-e  (w^(-e - 2048) = -w^-e)
    // base^(16 t): t < 256 -> exponent bits 4 .. 11
    Fr b16 = base;
#pragma unroll
    for (int k = 0; k < 4; k++) b16 = sqr(b16);
    Fr cur = ninv, pw = b16;
    for (uint32_t bits = t; bits; bits >>= 1) {
        if (bits & 1u) cur = cur * pw;
        pw = sqr(pw);
    }
    uint4 *out = coeffs_raw + ((size_t)blockIdx.x * kBlobElems + (size_t)t * 16) * 2;
#pragma unroll 1
    for (int k = 0; k < 16; k++) {
        uint32_t raw[8];
        fe_to_raw<FrParams>(raw, cur);
        out[2 * k] = make_uint4(raw[0], raw[1], raw[2], raw[3]);
        out[2 * k + 1] = make_uint4(raw[4], raw[5], raw[6], raw[7]);
        cur = cur * base;
    }
}

void launch_idft_columns(uint32_t *coeffs_raw, const Fr *tw_inv, uint32_t first, size_t n_rows, hipStream_t st) {
    ProfScope p("k_idft_columns", st);
    uint32_t n4096[8] = {4096, 0, 0, 0, 0, 0, 0, 0};
    const Fr ninv = inv(fe_from_raw<FrParams>(n4096));                    // 1 / 4096 (Montgomery form), on the host
    hipLaunchKernelGGL(k_idft_columns, dim3((unsigned)n_rows), dim3(256), 0, st, (uint4 *)coeffs_raw, tw_inv, first, ninv);
}

// c-kzg blobs on the Lagrange form: the canonical little-endian elements ARE the scalars the MSM reads (32-bit words, least
// significant first), so "parsing" is a copy with the range check of the c-kzg front end (an element >= r: status[blob] = BADARGS)
__global__ __launch_bounds__(256) void k_copy_le_check(const uint4 *__restrict__ blobs, uint4 *__restrict__ scalars_raw,
                                                       int32_t *__restrict__ status, size_t n_elems) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_elems) return;
    const uint4 lo = blobs[2 * i], hi = blobs[2 * i + 1];
    const uint32_t s[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    if (status && raw_geq<8>(s, FrParams::MOD)) status[i / kBlobElems] = kStatusBadArgs;
    scalars_raw[2 * i] = lo;
    scalars_raw[2 * i + 1] = hi;
}

void launch_copy_le_check(const uint8_t *blobs, uint32_t *scalars_raw, int32_t *status, size_t n_blobs, hipStream_t st) {
    ProfScope p("k_copy_le_check", st);
    const size_t n = n_blobs * kBlobElems;
    hipLaunchKernelGGL(k_copy_le_check, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const uint4 *)blobs, (uint4 *)scalars_raw, status, n);
}

// Quotients of a proof call whose MSM runs on the Lagrange form while the quotient was computed in coefficient form (a Lagrange-form
// table with no monomial table beside it): coefficients -> evaluations on the bit-reversed domain, i.e. a forward transform. Entry:
// canonical raw coefficient k -> Montgomery form at position bitrev(k) (the transform is decimation in time); exit: Montgomery
// evaluation at w^j -> canonical raw at position bitrev(j).
__global__ __launch_bounds__(256) void k_raw_to_mont_bitrev(const uint4 *__restrict__ in_raw, Fr *__restrict__ out, size_t n) {
    const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n) return;
    const uint32_t i = (uint32_t)(g & (kBlobElems - 1)), r = __brev(i) >> 20;
    const uint4 lo = in_raw[2 * g], hi = in_raw[2 * g + 1];
    const uint32_t s[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    out[(g - i) + r] = fe_from_raw<FrParams>(s);
}
__global__ __launch_bounds__(256) void k_mont_to_raw_bitrev(const Fr *__restrict__ in, uint4 *__restrict__ out_raw, size_t n) {
    const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n) return;
    const uint32_t i = (uint32_t)(g & (kBlobElems - 1)), r = __brev(i) >> 20;
    uint32_t s[8];
    fe_to_raw<FrParams>(s, in[g]);
    const size_t o = (g - i) + r;
    out_raw[2 * o] = make_uint4(s[0], s[1], s[2], s[3]);
    out_raw[2 * o + 1] = make_uint4(s[4], s[5], s[6], s[7]);
}

// coeffs_raw (in place) <- evaluations on the bit-reversed domain; `scratch`, `scratch2`: n_blobs x 4096 Fr each
void launch_coefficients_to_evaluations(uint32_t *coeffs_raw, Fr *scratch, Fr *scratch2, const Fr28 *tw28_fwd, size_t n_blobs, hipStream_t st) {
    const size_t n = n_blobs * kBlobElems;
    {
        ProfScope p("k_raw_to_mont_bitrev", st);
        hipLaunchKernelGGL(k_raw_to_mont_bitrev, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const uint4 *)coeffs_raw, scratch, n);
    }
    launch_ntt4096(scratch, scratch2, tw28_fwd, 0, n_blobs, st);
    {
        ProfScope p("k_mont_to_raw_bitrev", st);
        hipLaunchKernelGGL(k_mont_to_raw_bitrev, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const Fr *)scratch2, (uint4 *)coeffs_raw, n);
    }
}

// Montgomery Fr -> 32 bytes in the requested byte order, one lane each (z of a batch going back to the host)
__global__ __launch_bounds__(64) void k_fr_mont_to_bytes(const Fr *__restrict__ in, uint8_t *__restrict__ out, int le,
                                                         size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t s[8];
    fe_to_raw<FrParams>(s, in[i]);
    if (le) raw_to_le<8>(out + 32 * i, s); else raw_to_be<8>(out + 32 * i, s);
}
void launch_fr_mont_to_bytes(const Fr *in, uint8_t *out, int le, size_t n, hipStream_t st) {
    ProfScope p("k_fr_mont_to_bytes", st);
    hipLaunchKernelGGL(k_fr_mont_to_bytes, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, in, out, le, n);
}

__global__ __launch_bounds__(64) void k_z_from_bytes(const uint8_t *__restrict__ zb, Fr *__restrict__ z_mont,
                                                     int32_t *__restrict__ status, int le, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t s[8];
    if (le) {
        raw_from_le<8>(s, zb + 32 * i);
        if (status && raw_geq<8>(s, FrParams::MOD)) status[i] = kStatusBadArgs;  // status == NULL: reduce silently (digests)
    } else {
        raw_from_be<8>(s, zb + 32 * i);
    }
    z_mont[i] = fe_from_raw<FrParams>(s);  // reduces when >= r (reference mode)
}

void launch_z_from_bytes(const uint8_t *z_bytes, Fr *z_mont, int32_t *status, int le, size_t n, hipStream_t st) {
    ProfScope p("k_z_from_bytes", st);
    hipLaunchKernelGGL(k_z_from_bytes, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, z_bytes, z_mont, status, le, n);
}

}  // namespace lwk
