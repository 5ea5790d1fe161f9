// direct.hip -- "direct" fixed-base MSM: no buckets at all, paid for with HBM capacity.
//
// The MI355X has 288 GB of HBM3E and the SRS never changes. With every multiple d * 2^(C j) * P_i
// (d = 1 .. 2^(C-1)) of every setup point precomputed in affine form, a 4096-term MSM is nothing but the
// sum of one table row per (scalar, window): 4096 * ceil(255 / C) mixed additions -- 65,536 for C = 16,
// 69,632 for C = 15 -- with no digit sort, no bucket array and no bucket reduction (the bucket path of
// msm.hip spends 81,920 additions plus ~15 % on sort + reduction). The table is 4096 * 16 * 32768 * 112 B =
// 240 GB for C = 16, 135 GB for C = 15, 68 GB for C = 14 (8/7 of that with every row in a 128-byte line of its own,
// the layout used whenever it fits: kernels.h); its rows are gathered at random, 112 contiguous
// bytes per lane, which HBM sustains at 1.3e10 rows/s for tables of this size (tools/gather_bench.hip),
// twice what the arithmetic can consume. A load selects a 13 .. 10-bit table by itself (engine.hip:
// direct_from_env); the wider ones are opt-in (lwkzg_enable_direct_table).
//
// Replaces, like msm.hip, lambdaworks_math::msm::pippenger::msm as reached from KZG::commit / KZG::open
// (call sites /root/reference/src/lib.rs:242,270,329,394).
#include <stdlib.h>
#include <chrono>
#include "kernels.h"
#include "knobs.h"

namespace lwk {

template <int C>
struct DirectPlan {
    static constexpr int NW = (255 + C - 1) / C;          // windows; the top one is unsigned and takes the carry
    static constexpr int H = 1 << (C - 1);                // rows per (signed window, point): d = 1 .. 2^(C-1)
    static constexpr int WTOP = 255 - C * (NW - 1);       // bits in the top window (scalars are < r < 2^255)
    static constexpr int HTOP = 1 << WTOP;                // rows per (top window, point): d = 1 .. 2^WTOP
    static constexpr size_t TOP_BASE = (size_t)(NW - 1) * kBlobElems * H;
    static constexpr size_t ENTRIES = TOP_BASE + (size_t)kBlobElems * HTOP;
};

// the same plan as run-time values: the table builders and the generic accumulate kernel (widths 10 .. 13) take it as
// a kernel argument; the three widest widths keep compile-time plans in the hot loop
struct DirectPlanRt {
    int c, nw, wtop;
    uint32_t h, htop;
    size_t top_base, entries;
};

static DirectPlanRt make_plan(int bits) {
    DirectPlanRt p{};
    if (bits < kDirectMinBits || bits > kDirectMaxBits) return p;
    p.c = bits;
    p.nw = (255 + bits - 1) / bits;
    p.wtop = 255 - bits * (p.nw - 1);
    p.h = 1u << (bits - 1);
    p.htop = 1u << p.wtop;
    p.top_base = (size_t)(p.nw - 1) * kBlobElems * p.h;
    p.entries = p.top_base + (size_t)kBlobElems * p.htop;
    return p;
}

template <int C>
struct PlanOf {  // compile-time plan, converted
    __host__ __device__ static constexpr DirectPlanRt get() {
        return DirectPlanRt{C, DirectPlan<C>::NW, DirectPlan<C>::WTOP, (uint32_t)DirectPlan<C>::H, (uint32_t)DirectPlan<C>::HTOP,
                            DirectPlan<C>::TOP_BASE, DirectPlan<C>::ENTRIES};
    }
};

size_t direct_table_entries(int bits) { return make_plan(bits).entries; }
int direct_num_windows(int bits) { return make_plan(bits).nw; }

// ------------------------------------------------------------------------------------------------
// table build, step 1: Q[j][i] = 2^(C j) P_i in affine hot-loop form (one lane per point)

__global__ __launch_bounds__(64) void k_direct_qbase(const G1Affine *__restrict__ points, G1Affine29 *__restrict__ qbase, int C,
                                                     int NW) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= kBlobElems) return;
    G1Affine29 a = affine_to_29(points[i]);
    qbase[i] = a;
    G1Xyzz29 cur = G1Xyzz29::from_affine(a.x, a.y);
    for (int j = 1; j < NW; j++) {
        for (int d = 0; d < C; d++) cur = xyzz_dbl(cur);
        qbase[(size_t)j * kBlobElems + i] = xyzz29_to_affine29(cur);  // never infinity: P has prime order r
    }
}

// ------------------------------------------------------------------------------------------------
// table build, step 2: all multiples. One lane per chunk of kChunk consecutive multiples of one Q:
//   forward  : walk (s0 + m) Q in XYZZ; park each point and the running product of its ZZ*ZZZ in lane-private
//              scratch (five 56-byte values per row, interleaved across lanes so the accesses coalesce)
//   invert   : ONE field inversion per chunk (Montgomery's trick)
//   backward : peel the individual inverses off the running product and emit the affine rows.
// About 25 field products per row; the scratch traffic (560 B per row) is what the build time is made of.

constexpr int kChunk = 64;

// M = rows per (window, point) pair in this launch (a power of two)
__global__ __launch_bounds__(256) void k_direct_build(const G1Affine29 *__restrict__ qbase_win0, G1Affine29 *__restrict__ out_win0,
                                                      size_t n_pairs, F29<2> *__restrict__ scratch, size_t n_threads, int M,
                                                      size_t row_bytes) {
    const int K = M < kChunk ? M : kChunk;
    const size_t kChunksPerPair = (size_t)(M / K);
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t n_work = n_pairs * kChunksPerPair;
    // scratch rows: [m][5][lane] -> X, Y, ZZ, ZZZ (weakly reduced to < 2p) and the prefix product
    auto slot = [&](int m, int f) -> F29<2> & { return scratch[((size_t)m * 5 + f) * n_threads + gid]; };
    for (size_t w = gid; w < n_work; w += n_threads) {
        const size_t pair = w / kChunksPerPair;
        const uint32_t s0 = (uint32_t)(w % kChunksPerPair) * K + 1;
        const G1Affine29 q = qbase_win0[pair];
        // cur = [s0] Q
        G1Xyzz29 cur = G1Xyzz29::infinity();
        for (int bit = 16; bit >= 0; bit--) {
            cur = xyzz_dbl(cur);
            if ((s0 >> bit) & 1) cur = xyzz_madd(cur, q.x, q.y);
        }
        F29<2> pref = F29<2>::one();
        const F29<1> one = F29<1>::one();
        for (int m = 0; m < K; m++) {
            F29<2> t = cur.zz * cur.zzz;
            pref = (m == 0) ? t : F29<2>(pref * t);
            slot(m, 0) = cur.x * one;  // X < 14p -> < 2p (same residue)
            slot(m, 1) = cur.y * one;
            slot(m, 2) = cur.zz;
            slot(m, 3) = cur.zzz;
            slot(m, 4) = pref;
            if (m < K - 1) cur = xyzz_madd(cur, q.x, q.y);
        }
        F29<2> inv = f29_inv(pref);
        char *rows = (char *)out_win0 + (pair * (size_t)M + (s0 - 1)) * row_bytes;
        for (int m = K - 1; m >= 0; m--) {
            F29<2> zz = slot(m, 2), zzz = slot(m, 3);
            F29<2> tinv = (m > 0) ? F29<2>(inv * slot(m - 1, 4)) : inv;  // 1 / (ZZ ZZZ) of row m
            inv = inv * (zz * zzz);                                      // inverse of the prefix product up to row m - 1
            G1Affine29 r;
            r.x = slot(m, 0) * (tinv * zzz);  // X / ZZ
            r.y = slot(m, 1) * (tinv * zz);   // Y / ZZZ
            *(G1Affine29 *)(rows + (size_t)m * row_bytes) = r;
        }
    }
}

static double wall_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

void free_direct_table(DirectTable &t) {
    for (int j = 0; j < kDirectMaxWindows; j++)
        if (t.win[j]) {
            hipFree(t.win[j]);
            t.win[j] = nullptr;
        }
    if (t.win_dev) hipFree(t.win_dev);
    t.win_dev = nullptr;
    t.nw = 0;
    t.bytes = 0;
}

// in_place: `t` already holds a table of the same width and row size (over ANOTHER point set: the other form of the setup) -- its
// window allocations are kept and only the build kernels run again over them: no hipFree, no hipMalloc, no wait for the driver's
// scrub of 275 GB just released (lwkzg_settings_set_mode on a table that leaves no room for a second one: 7-8.5 s -> the kernels' 0.5 s)
hipError_t build_direct_table(int bits, const G1Affine *points, DirectTable &t, size_t row_bytes, hipStream_t st, double *ms, bool in_place) {
    const DirectPlanRt P = make_plan(bits);
    if (!P.entries || P.nw > kDirectMaxWindows) return hipErrorInvalidValue;
    if (in_place && (t.nw != P.nw || t.bytes != P.entries * row_bytes || !t.win_dev)) return hipErrorInvalidValue;
    if (!in_place) free_direct_table(t);
    G1Affine29 *qbase = nullptr;
    F29<2> *scratch = nullptr;
    const double t0 = wall_ms();
    // lanes in flight during the build: 4.7 GB of scratch for the wide tables, a quarter of that (one wave per SIMD) for
    // the narrow ones (the default engine's table should not need gigabytes of headroom to be built)
    const size_t n_threads = bits >= 14 ? 256 * 1024 : 64 * 1024;
    hipError_t e = hipMalloc((void **)&qbase, (size_t)P.nw * kBlobElems * sizeof(G1Affine29));
    if (e == hipSuccess) e = hipMalloc((void **)&scratch, (size_t)kChunk * 5 * n_threads * sizeof(F29<2>));
    if (e == hipSuccess && !in_place) e = hipMalloc((void **)&t.win_dev, kDirectMaxWindows * sizeof(uint64_t));
    const double t1 = wall_ms();
    double malloc_ms = 0, t_last_malloc = t1;
    if (e == hipSuccess) {
        {
            ProfScope p("k_direct_qbase", st);
            hipLaunchKernelGGL(k_direct_qbase, dim3(kBlobElems / 64), dim3(64), 0, st, points, qbase, P.c, P.nw);
        }
        // window by window: the host allocates window j + 1 while the GPU builds window j (the launches are asynchronous)
        for (int j = 0; j < P.nw && e == hipSuccess; j++) {
            const bool top = j == P.nw - 1;
            const size_t rows = (size_t)kBlobElems * (top ? P.htop : P.h);
            if (!in_place) {
                const double a0 = wall_ms();
                e = hipMalloc(&t.win[j], rows * row_bytes);
                t_last_malloc = wall_ms();
                malloc_ms += t_last_malloc - a0;
                if (e != hipSuccess) break;
                t.bytes += rows * row_bytes;
            }
            ProfScope p("k_direct_build", st);
            hipLaunchKernelGGL(k_direct_build, dim3((unsigned)(n_threads / 256)), dim3(256), 0, st, qbase + (size_t)j * kBlobElems,
                               (G1Affine29 *)t.win[j], (size_t)kBlobElems, scratch, n_threads, (int)(top ? P.htop : P.h), row_bytes);
        }
        if (e == hipSuccess && !in_place) {
            uint64_t h[kDirectMaxWindows] = {};
            for (int j = 0; j < P.nw; j++) h[j] = (uint64_t)(uintptr_t)t.win[j];
            e = hipMemcpyAsync(t.win_dev, h, sizeof h, hipMemcpyHostToDevice, st);
        }
        const hipError_t es = hipStreamSynchronize(st);  // (also on failure: the builds of the windows that exist must be over before they are freed)
        if (e == hipSuccess) e = es;
    }
    const double t2 = wall_ms();
    if (qbase) hipFree(qbase);
    if (scratch) hipFree(scratch);
    if (e != hipSuccess) {
        free_direct_table(t);
    } else {
        t.nw = P.nw;
    }
    if (ms) {
        ms[0] = t1 - t0;
        ms[1] = malloc_ms;
        ms[2] = t2 - t_last_malloc;
        ms[3] = wall_ms() - t2;
    }
    return e;
}

// ------------------------------------------------------------------------------------------------
// the MSM: every lane owns a strided set of scalars of one blob, turns each into <= NW table rows on the fly
// (signed C-bit digits, top window unsigned), gathers the row while the previous mixed addition runs, and keeps
// its partial sum in VGPRs; the workgroup folds its 256 partial sums by wave shuffles + one LDS hop.

#ifndef LWK_DIR_THREADS
#define LWK_DIR_THREADS 256
#endif
constexpr int kDirThreads = LWK_DIR_THREADS;

// 64 lanes -> lane 0, by shuffles (no LDS). The six additions are a serial chain that every workgroup ends with, and
// for a handful of blobs they ARE the run time: field products inlined here as in the accumulate loop.
__device__ __forceinline__ G1Xyzz29 wave_fold(const G1Xyzz29 &in, int lane, int width = 64) {
    G1Xyzz29i s = *(const G1Xyzz29i *)&in;
    for (int d = width / 2; d >= 1; d >>= 1) {
        G1Xyzz29i other;
#pragma unroll
        for (int k = 0; k < 14; k++) {
            other.x.l[k] = __shfl_down(s.x.l[k], d, 64);
            other.y.l[k] = __shfl_down(s.y.l[k], d, 64);
            other.zz.l[k] = __shfl_down(s.zz.l[k], d, 64);
            other.zzz.l[k] = __shfl_down(s.zzz.l[k], d, 64);
        }
        if (lane < d) s = xyzz_add(s, other);
    }
    return *(G1Xyzz29 *)&s;
}

// CT = the window width as a compile-time constant (14, 15, 16), or 0: the plan is the kernel argument `rt`
template <int CT>
__global__ __launch_bounds__(kDirThreads) void k_direct_accumulate(const uint64_t *__restrict__ win_dev,
                                                                   const uint4 *__restrict__ scalars,
                                                                   G1Xyzz29 *__restrict__ partials, int scalars_per_lane,
                                                                   DirectPlanRt rt, uint32_t row_bytes,
                                                                   const uint32_t *__restrict__ redo) {
    // second pass behind the hand-scheduled kernel (k_direct_accumulate_asm): only the blobs it flagged are recomputed
    if (redo && !redo[blockIdx.y]) return;
    const DirectPlanRt P = CT ? PlanOf<CT ? CT : 16>::get() : rt;  // folds to constants when CT != 0
    const int C = P.c;
    __shared__ uint32_t limbs[8 * kDirThreads];   // the lane's current scalar, for run-time window indexing
    __shared__ G1Xyzz29 wave_sum[kDirThreads / 64];
    __shared__ uint64_t win_base[kDirectMaxWindows];  // the table is one allocation per window (kernels.h: DirectTable)
    if (threadIdx.x < kDirectMaxWindows) win_base[threadIdx.x] = win_dev[threadIdx.x];
    __syncthreads();
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t blob = blockIdx.y;
    const int lanes_per_blob = gridDim.x * kDirThreads;
    const int first = blockIdx.x * kDirThreads + tid;
    const uint4 *sc = scalars + blob * (size_t)kBlobElems * 2;
    // tiny batches also split the windows of a scalar over gridDim.z workgroups; the signed-digit carry chain is
    // still walked from window 0 (a few integer ops per window), only the gathers and additions are confined
    const int wper = (P.nw + (int)gridDim.z - 1) / (int)gridDim.z;
    const int w_lo = (int)blockIdx.z * wper, w_hi = w_lo + wper;

    int q = 0, j = 0;
    uint32_t carry = 0, point = 0;
    // next non-zero digit of this lane's scalar stream: row address resolved and its gather issued
    auto advance = [&](bool &valid, uint32_t &neg, G1Affine29i &row) {
        valid = false;
        while (q < scalars_per_lane) {
            if (j == 0) {
                point = (uint32_t)(first + q * lanes_per_blob);
                uint4 lo = sc[2 * point], hi = sc[2 * point + 1];
                limbs[0 * kDirThreads + tid] = lo.x; limbs[1 * kDirThreads + tid] = lo.y;
                limbs[2 * kDirThreads + tid] = lo.z; limbs[3 * kDirThreads + tid] = lo.w;
                limbs[4 * kDirThreads + tid] = hi.x; limbs[5 * kDirThreads + tid] = hi.y;
                limbs[6 * kDirThreads + tid] = hi.z; limbs[7 * kDirThreads + tid] = hi.w;
                carry = 0;
            }
            const int o = j * C, limb = o >> 5, sh = o & 31;
            uint32_t v = limbs[limb * kDirThreads + tid] >> sh;
            if (sh + C > 32 && limb + 1 < 8) v |= limbs[(limb + 1) * kDirThreads + tid] << (32 - sh);
            const bool top = j == P.nw - 1;
            uint32_t raw = (v & ((top ? (1u << P.wtop) : (1u << C)) - 1u)) + carry;
            uint32_t ng = (!top && raw > P.h) ? 1u : 0u;
            uint32_t mag = ng ? (1u << C) - raw : raw;
            carry = ng;
            const char *wbase = (const char *)(uintptr_t)win_base[j];
            const size_t idx = (size_t)point * (top ? P.htop : P.h) + (mag - 1);  // row within window j
            j++;
            if (j == P.nw) {
                j = 0;
                q++;
            }
            const int jw = (j == 0 ? P.nw : j) - 1;  // the window this digit belongs to (j was advanced above)
            if (mag && jw >= w_lo && jw < w_hi) {
                valid = true;
                neg = ng;
                // (plain loads: with the nt hint the seven loads of a row no longer meet in the cache, -7 %)
                row = *(const G1Affine29i *)(wbase + idx * row_bytes);
                return;
            }
        }
    };

    // the next row's gather is issued inside the addition, right after the current row's last use (xyzz_madd_split)
    G1Xyzz29i acc = G1Xyzz29i::infinity();
    bool cv;
    uint32_t cn;
    G1Affine29i cr;
    advance(cv, cn, cr);
    while (cv) {
        const auto qy = cneg(cr.y, cn != 0);
        xyzz_madd_split(acc, cr.x, qy, [&]() { advance(cv, cn, cr); });
    }

    // fold: 64 lanes by shuffles, 4 waves through LDS
    G1Xyzz29 s = wave_fold(*(G1Xyzz29 *)&acc, lane);
    if (lane == 0) wave_sum[wave] = s;
    __syncthreads();
    if (wave == 0) {  // the four wave sums: two more shuffle levels on the first wave
        G1Xyzz29 t = lane < kDirThreads / 64 ? wave_sum[lane] : G1Xyzz29::infinity();
        t = wave_fold(t, lane, kDirThreads / 64);
        if (lane == 0) partials[(blob * gridDim.z + blockIdx.z) * gridDim.x + blockIdx.x] = t;
    }
}

// ------------------------------------------------------------------------------------------------
// The same accumulation as a hand-scheduled instruction stream (tools/gen_direct_asm.py writes direct_asm.inc and
// explains what it does differently). The statement is the whole kernel: it reads its operands,
// runs the loop over (scalar, window) with every register placed by the generator, and stores each LANE's partial sum
// (-X, -Y, ZZ, ZZZ: 56 words, literal zeros for a lane that added nothing) to lane_out[blob][block][lane];
// k_direct_fold_lanes adds the 256 lanes of a workgroup. A lane that meets P = +-Q (or would have to double) sets
// redo[blob]; k_direct_accumulate<CT> then recomputes exactly those blobs with its complete branches.
constexpr int kLaneWords = 56;

__global__ __launch_bounds__(kDirThreads) void k_direct_accumulate_asm(const uint64_t *__restrict__ win_dev,
                                                                       const uint4 *__restrict__ scalars,
                                                                       uint32_t *__restrict__ lane_out, uint32_t *__restrict__ redo,
                                                                       int scalars_per_lane, DirectPlanRt rt, uint32_t row_bytes) {
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t tid = threadIdx.x;
    const uint32_t first = blockIdx.x * blockDim.x + tid;    // (workgroups of 128 or 256 lanes: launch_direct_t)
    const uint4 *sc = scalars + (size_t)blockIdx.y * kBlobElems * 2;
    uint32_t *out = lane_out + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * (size_t)(blockDim.x * kLaneWords);
    uint32_t *flag = redo + blockIdx.y;
    const uint32_t lanes_per_blob = gridDim.x * blockDim.x;
    asm volatile(
#include "direct_asm.inc"
        :
        : "s"(win_dev), "s"(sc), "s"(out), "s"(flag), "s"(scalars_per_lane), "s"(lanes_per_blob), "s"(rt.c), "s"(rt.nw), "s"(rt.wtop),
          "s"(rt.h), "s"(rt.htop), "s"(row_bytes), "v"(first), "v"(tid)
        :
#include "direct_asm_clobbers.inc"
    );
#endif
}

// the 256 lane sums of one workgroup of k_direct_accumulate_asm -> one partial sum (a "unit"). ONE wave per unit: every thread
// first adds four lane sums in sequence (all 64 lanes busy), then six shuffle levels -- nine additions deep on a quarter of
// the waves, where a 256-thread tree is eight deep with most lanes idle on four times as many (0.48 -> 0.30 ms at 1024 blobs).
// This is the compiler-scheduled arm (LWKZG_FOLD_ASM=0); k_direct_fold_lanes_asm below is what runs.

// LWK_FOLD_CALL: field products of the lane fold as calls of the shared product function (small code) instead of inlined
#ifdef LWK_FOLD_CALL
constexpr bool kFoldInl = false;
typedef G1Xyzz29 FoldPoint;
#else
constexpr bool kFoldInl = true;
typedef G1Xyzz29i FoldPoint;
#endif

__device__ __forceinline__ FoldPoint load_lane_sum(const uint32_t *src) {
    // bounds as tools/gen_direct_asm.py leaves them: -X < 25p, and -Y, ZZ, ZZZ < 17p (products whose reduction digits are
    // not masked); one product by 1 each brings them to the < 2p the point type holds
    F29<25, kFoldInl, 1> nx;
    F29<17, kFoldInl, 1> ny, zz, zzz;
    FoldPoint acc;
    const uint4 *s4 = (const uint4 *)src;
    uint32_t wds[kLaneWords];
#pragma unroll
    for (int k = 0; k < kLaneWords / 4; k++) {
        uint4 t = s4[k];
        wds[4 * k] = t.x; wds[4 * k + 1] = t.y; wds[4 * k + 2] = t.z; wds[4 * k + 3] = t.w;
    }
#pragma unroll
    for (int i = 0; i < 14; i++) {
        nx.l[i] = wds[i];
        ny.l[i] = wds[14 + i];
        zz.l[i] = wds[28 + i];
        zzz.l[i] = wds[42 + i];
    }
    const F29<1, kFoldInl> one = F29<1, kFoldInl>::one();
    const bool inf = zz.is_literal_zero();
    acc.x = neg(nx) * one;  // the stream keeps -X and -Y
    acc.y = neg(ny) * one;
    acc.zz = zz * one;
    acc.zzz = zzz * one;
    if (inf) acc = FoldPoint::infinity();
    return acc;
}

__global__ __launch_bounds__(64) void k_direct_fold_lanes(const uint32_t *__restrict__ lane_out, G1Xyzz29 *__restrict__ partials,
                                                          const uint32_t *__restrict__ redo, int lanes_per_block, int blocks_per_blob) {
    const int unit = blockIdx.x;  // = blob * blocks_per_blob + block
    if (redo[unit / blocks_per_blob]) return;  // recomputed by the second pass
    const int kFoldPerThread = lanes_per_block / 64;
    __builtin_amdgcn_s_setprio(2);
    const int lane = threadIdx.x;
    const uint32_t *src = lane_out + ((size_t)unit * lanes_per_block + lane) * (size_t)kLaneWords;
    FoldPoint acc = load_lane_sum(src);
    // ONE call site of the (inlined, 48 KB) addition for all nine steps, so that the kernel stays inside the instruction cache:
    // steps 0 .. 2 add this thread's other three lane sums, steps 3 .. 8 are the shuffle tree
#pragma unroll 1
    for (int step = 1; step < kFoldPerThread + 6; step++) {
        FoldPoint other;
        bool take = true;
        if (step < kFoldPerThread) {
            other = load_lane_sum(src + (size_t)step * 64 * kLaneWords);
        } else {
            const int d = 32 >> (step - kFoldPerThread);
#pragma unroll
            for (int k = 0; k < 14; k++) {
                other.x.l[k] = __shfl_down(acc.x.l[k], d, 64);
                other.y.l[k] = __shfl_down(acc.y.l[k], d, 64);
                other.zz.l[k] = __shfl_down(acc.zz.l[k], d, 64);
                other.zzz.l[k] = __shfl_down(acc.zzz.l[k], d, 64);
            }
            take = lane < d;
        }
        if (take) acc = xyzz_add(acc, other);
    }
    if (lane == 0) partials[unit] = *(G1Xyzz29 *)&acc;
}

// The same fold as a hand-scheduled stream (tools/gen_fold_asm.py, direct_fold_asm.inc): one wave per unit, the additions'
// independent products three chains at a time; LWKZG_FOLD_ASM=0 selects the compiler's schedule above (the A/B arm). A pair of
// equal or opposite sums raises the blob's redo flag. Two builds of the one stream: ALONE lists 64 accumulator registers it never
// touches as clobbers, which takes the kernel past 256 registers per lane, i.e. ONE wave per SIMD -- for launches of a whole
// chip's worth of units (>= 1024), where the dispatcher otherwise puts two of these dependent chains on some SIMDs and none on
// others and the pairs set the duration (0.245 -> 0.208 ms; profiles/r03_experiments.md section 6). Smaller launches run beside
// other streams' kernels and keep the shareable build (the host-pointer slices lost 10 % to the exclusive one).
__global__ __launch_bounds__(64) void k_direct_fold_lanes_asm(const uint32_t *__restrict__ lane_out, G1Xyzz29 *__restrict__ partials,
                                                              uint32_t *__restrict__ redo, int lanes_per_block, int blocks_per_blob) {
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t unit = blockIdx.x;  // = blob * blocks_per_blob + block
    uint32_t *flag = redo + unit / blocks_per_blob;
    if (*flag) return;  // recomputed by the second pass
    const uint32_t *src = lane_out + (size_t)unit * lanes_per_block * kLaneWords;
    G1Xyzz29 *out = partials + unit;
    const uint32_t per = lanes_per_block / 64, lane = threadIdx.x, levels = 6;
    asm volatile(
#include "direct_fold_asm.inc"
        :
        : "s"(src), "s"(out), "s"(flag), "s"(per), "v"(lane), "s"(levels)
        :
#include "direct_fold_asm_clobbers.inc"
    );
#endif
}

__global__ __launch_bounds__(64) void k_direct_fold_lanes_asm_alone(const uint32_t *__restrict__ lane_out, G1Xyzz29 *__restrict__ partials,
                                                                    uint32_t *__restrict__ redo, int lanes_per_block, int blocks_per_blob) {
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t unit = blockIdx.x;
    uint32_t *flag = redo + unit / blocks_per_blob;
    if (*flag) return;
    const uint32_t *src = lane_out + (size_t)unit * lanes_per_block * kLaneWords;
    G1Xyzz29 *out = partials + unit;
    const uint32_t per = lanes_per_block / 64, lane = threadIdx.x, levels = 6;
    asm volatile(
#include "direct_fold_asm.inc"
        :
        : "s"(src), "s"(out), "s"(flag), "s"(per), "v"(lane), "s"(levels)
        :
#include "direct_fold_asm_clobbers_pad.inc"
    );
#endif
}

// sums[blob] = sum of its per-block partial sums (only when a blob was spread over several workgroups):
// one wave per blob, a shuffle tree over the <= 64 partials
__global__ __launch_bounds__(64) void k_direct_fold(const G1Xyzz29 *__restrict__ partials, G1Xyzz29 *__restrict__ sums,
                                                    int per_blob) {
    const size_t b = blockIdx.x;
    const int lane = threadIdx.x;
    __builtin_amdgcn_s_setprio(2);  // a short latency chain: it should not queue behind another call's MSM waves (engine.hip: pick_ctx)
    G1Xyzz29 s = lane < per_blob ? partials[b * per_blob + lane] : G1Xyzz29::infinity();
    s = wave_fold(s, lane);
    if (lane == 0) sums[b] = s;
}

// ------------------------------------------------------------------------------------------------
// A handful of blobs: the cooperative kernel (tools/gen_coop_asm.py writes coop_asm.inc and explains the design; DESIGN.md
// section 4d). One blob is a reduction tree of 16 levels on a chip of 1024 SIMDs, and a lone wave issues one instruction per four
// cycles whatever it depends on, so the way to a short call is more LANES per addition: a quad of lanes owns a point (lane c =
// coordinate c of X, Y, ZZ, ZZZ) and the addition's products run four at a time. Workgroups are single waves of 16 quads; a quad
// adds `rpq` rows of one scalar, the wave folds its quads by ds_bpermute, and waves hand their sums on through memory, sixteen to
// one, the last arrival of a group carrying on (no wave ever waits). The blob's sum lands in sums[blob] in the library's layout;
// a P = +-Q inside the formulas raises redo[blob] and the complete-branches kernel above recomputes that blob.
struct CoopParams {
    uint32_t k[8];        // the recoding constant: digit_j = window_j(scalar + K) - (H - 1)
    uint32_t pack;        // c | nw << 8 | rows per quad << 16 | log2(points) << 24
    uint32_t wtop, row_bytes, n0;
    uint32_t part_words, ctr_words;   // this blob's slice of the hand-off memory
};

constexpr uint32_t kCoopUnitWords = 56;   // a partial sum: 224 bytes

// waves of one blob at stage 0, the partial sums and counters all its stages need
static void coop_geometry(const DirectPlanRt &P, int rpq, uint32_t &n0, uint32_t &units, uint32_t &counters) {
    const uint32_t groups = (uint32_t)((P.nw + rpq - 1) / rpq);
    n0 = kBlobElems * groups / 16;
    units = 0;
    counters = 0;
    for (uint32_t n = n0; n > 1; n = (n + 15) / 16) {
        units += n;
        counters += (n + 15) / 16;
    }
}

// (workgroups of four waves that have nothing to do with each other: the dispatcher spreads a workgroup's waves over the four SIMDs
// of its compute unit, so 256 workgroups per blob are exactly one wave per SIMD, where single-wave workgroups pile up two and three
// to a SIMD and leave others empty)
__global__ __launch_bounds__(256) void k_coop_msm_asm(const uint64_t *__restrict__ win_dev, const uint4 *__restrict__ scalars,
                                                     uint32_t *__restrict__ partials, uint32_t *__restrict__ counters,
                                                     G1Xyzz29 *__restrict__ sums, uint32_t *__restrict__ redo, CoopParams prm) {
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t blob = blockIdx.y, lane = threadIdx.x & 63;
    const uint32_t unit = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint4 *sc = scalars + (size_t)blob * kBlobElems * 2;
    uint32_t *part = partials + (size_t)blob * prm.part_words;
    uint32_t *ctr = counters + (size_t)blob * prm.ctr_words;
    G1Xyzz29 *out = sums + blob;
    uint32_t *flag = redo + blob;
    asm volatile(
#include "coop_asm.inc"
        :
        : "s"(win_dev), "s"(sc), "s"(part), "s"(ctr), "s"(out), "s"(flag), "s"(prm.k[0]), "s"(prm.k[1]), "s"(prm.k[2]), "s"(prm.k[3]),
          "s"(prm.k[4]), "s"(prm.k[5]), "s"(prm.k[6]), "s"(prm.k[7]), "s"(prm.pack), "s"(prm.wtop), "s"(prm.row_bytes), "s"(prm.n0), "s"(unit),
          "v"(lane)
        :
#include "coop_asm_clobbers.inc"
    );
#endif
}

// LWKZG_COOP=0 switches the cooperative kernel off; LWKZG_COOP_MAX = the largest batch it takes (default 8);
static int coop_max_blobs() { return knobs().coop == 0 ? 0 : knobs().coop_max; }   // (experiment knobs; coop_max is clamped to 1..8)
// rows a quad adds before the tree. A blob is 256 waves per group of `rpq` windows; the chip has 1024 SIMDs and a wave that shares
// its SIMD takes twice as long, so the launch aims at 1024 waves in all: four window groups for one blob (16 windows: 4 rows per
// quad, 20: 5), two for two blobs, one from four blobs on. LWKZG_COOP_RPQ overrides (experiments).
static int coop_rows_per_quad(int nw, size_t n_blobs) {
    const int env = knobs().coop_rpq < 0 ? 0 : knobs().coop_rpq > 32 ? 32 : knobs().coop_rpq;
    if (env) return env;
    const int groups = n_blobs >= 4 ? 1 : (int)(4 / n_blobs);
    return (nw + groups - 1) / groups;
}

// LWKZG_DIRECT_ASM=0 keeps every launch on the compiler-scheduled kernel (the A/B arm)
static bool direct_asm_enabled() { return knobs().direct_asm; }

template <int CT>
static void launch_direct_t(const DirectPlanRt &plan, const uint64_t *table, size_t row_bytes, const uint32_t *scalars_raw,
                            G1Xyzz29 *lane_scratch, G1Xyzz29 *partials, uint32_t *redo, G1Xyzz29 *sums, size_t n_blobs, hipStream_t st,
                            int fill, uint32_t *redo_flag_out) {
    // many blobs: one workgroup per blob (16 scalars per lane, fewest fold steps); few blobs: spread each over up to
    // 16 workgroups so the chip fills and the dependent chain per lane stays short. `fill` = the number of workgroups
    // to aim for: 512 (two per compute unit, one round) when the kernel has the chip alone; 2048 when the settings
    // object runs two overlapping pipelines (engine.hip: pick_ctx), where workgroups of a quarter of the length keep the
    // tail short although another call's waves take compute-unit slots away (60k instead of 53k proofs/s at 256 blobs)
    const int kFillEnv = knobs().direct_fill;
    const int kFill = kFillEnv ? kFillEnv : fill ? fill : 512;
    // the cooperative kernel's hand-off lives in workspace that is idle on this path: a blob's sums in its slice of `lane_scratch` (4096
    // XYZZ per blob), its counters in the words behind the redo flags (kNumBuckets + 1 per blob). Geometries that do not fit -- only
    // reachable through the experiment knob LWKZG_COOP_RPQ (1 row per quad on a 16- or 20-window table: 4369 / 5461 units) -- take the
    // throughput kernels instead of writing past the slice
    CoopParams prm{};
    const int rpq = coop_rows_per_quad(plan.nw, n_blobs ? n_blobs : 1);
    uint32_t units = 0, counters = 0;
    coop_geometry(plan, rpq, prm.n0, units, counters);
    const bool coop_fits = units <= (uint32_t)kBlobElems && 1 + counters <= (uint32_t)kNumBuckets + 1;
    if ((int)n_blobs <= coop_max_blobs() && !fill && coop_fits) {
        // `lane_scratch` takes the hand-off sums, the words behind the redo flags the hand-off counters
        // K = (H - 1) * sum of 2^(C j) over the signed windows (all but the top one)
        for (int j = 0; j + 1 < plan.nw; j++) {
            const int bit = plan.c * j;
            const uint64_t v = (uint64_t)(plan.h - 1) << (bit & 31);
            uint64_t carry = v;
            for (int w = bit >> 5; w < 8 && carry; w++) {
                carry += prm.k[w];
                prm.k[w] = (uint32_t)carry;
                carry >>= 32;
            }
        }
        prm.pack = (uint32_t)plan.c | ((uint32_t)plan.nw << 8) | ((uint32_t)rpq << 16) | (12u << 24);
        prm.wtop = (uint32_t)plan.wtop;
        prm.row_bytes = (uint32_t)row_bytes;
        prm.part_words = units * kCoopUnitWords;
        prm.ctr_words = counters;
        uint32_t *ctr = redo + n_blobs;
        if (redo_flag_out) {
            // a one-blob call (r06, engine.hip: combine_run): the hand-off counters have been cleared by the parse kernel in front of this
            // launch, the redo flag is a word of pinned host memory the caller cleared and will look at after its one synchronisation -- no
            // fill launch, and no second-pass launch that would exit at its first instruction (the caller repeats a flagged call the long way)
            ProfScope p("k_coop_msm_asm", st);
            hipLaunchKernelGGL(k_coop_msm_asm, dim3(prm.n0 / 4, (unsigned)n_blobs), dim3(256), 0, st, table, (const uint4 *)scalars_raw,
                               (uint32_t *)lane_scratch, ctr, sums, redo_flag_out, prm);
            return;
        }
        hipMemsetAsync(redo, 0, n_blobs * (1 + (size_t)counters) * sizeof(uint32_t), st);
        {
            ProfScope p("k_coop_msm_asm", st);
            hipLaunchKernelGGL(k_coop_msm_asm, dim3(prm.n0 / 4, (unsigned)n_blobs), dim3(256), 0, st, table, (const uint4 *)scalars_raw,
                               (uint32_t *)lane_scratch, ctr, sums, redo, prm);
        }
        {
            // (one workgroup per flagged blob: slow and complete; none on honest data)
            ProfScope p("k_direct_redo", st);
            hipLaunchKernelGGL(k_direct_accumulate<CT>, dim3(1, (unsigned)n_blobs, 1), dim3(kDirThreads), 0, st, table,
                               (const uint4 *)scalars_raw, sums, kBlobElems / kDirThreads, plan, (uint32_t)row_bytes, (const uint32_t *)redo);
        }
        return;
    }
    int blocks_per_blob = 1, wsplit = 1;
    while (blocks_per_blob < 16 && n_blobs * blocks_per_blob < (size_t)kFill) blocks_per_blob <<= 1;
    while (wsplit < 4 && n_blobs * blocks_per_blob * wsplit * 8 <= (size_t)kFill) wsplit <<= 1;  // only for a handful of blobs
    const int scalars_per_lane = kBlobElems / (kDirThreads * blocks_per_blob);
    const int parts = blocks_per_blob * wsplit;
    G1Xyzz29 *dest = parts == 1 ? sums : partials;
    if (direct_asm_enabled() && wsplit == 1) {
        // the hand-scheduled stream, its lane fold, and a second pass of the C++ kernel over the blobs it flagged (none on
        // honest data: the launch exits at its first instruction). (Workgroups of 128 lanes -- one round of workgroups instead
        // of two, 128 lane sums per blob to fold instead of 256 -- measured 0.9 % slower in the accumulation than they save
        // in the fold: profiles/r03_experiments.md.)
        const int threads = kDirThreads;
        hipMemsetAsync(redo, 0, n_blobs * sizeof(uint32_t), st);
        {
            ProfScope p("k_direct_accumulate_asm", st);
            hipLaunchKernelGGL(k_direct_accumulate_asm, dim3(blocks_per_blob, (unsigned)n_blobs), dim3(threads), 0, st, table,
                               (const uint4 *)scalars_raw, (uint32_t *)lane_scratch, redo, kBlobElems / (threads * blocks_per_blob), plan,
                               (uint32_t)row_bytes);
        }
        {
            ProfScope p("k_direct_fold_lanes", st);
            const int n_units = (int)n_blobs * blocks_per_blob;
            const bool fold_asm = knobs().fold_asm;
            if (fold_asm) {
                if (n_units >= 1024 && !fill)   // a chip's worth of units and nobody beside us: one wave per SIMD
                    hipLaunchKernelGGL(k_direct_fold_lanes_asm_alone, dim3((unsigned)n_units), dim3(64), 0, st, (const uint32_t *)lane_scratch,
                                       dest, redo, threads, blocks_per_blob);
                else
                    hipLaunchKernelGGL(k_direct_fold_lanes_asm, dim3((unsigned)n_units), dim3(64), 0, st, (const uint32_t *)lane_scratch, dest,
                                       redo, threads, blocks_per_blob);
            } else {
                hipLaunchKernelGGL(k_direct_fold_lanes, dim3((unsigned)n_units), dim3(64), 0, st, (const uint32_t *)lane_scratch, dest,
                                   (const uint32_t *)redo, threads, blocks_per_blob);
            }
        }
        {
            ProfScope p("k_direct_redo", st);
            hipLaunchKernelGGL(k_direct_accumulate<CT>, dim3(blocks_per_blob, (unsigned)n_blobs, 1), dim3(kDirThreads), 0, st, table,
                               (const uint4 *)scalars_raw, dest, scalars_per_lane, plan, (uint32_t)row_bytes, (const uint32_t *)redo);
        }
    } else {
        ProfScope p("k_direct_accumulate", st);
        hipLaunchKernelGGL(k_direct_accumulate<CT>, dim3(blocks_per_blob, (unsigned)n_blobs, wsplit), dim3(kDirThreads), 0, st,
                           table, (const uint4 *)scalars_raw, dest, scalars_per_lane, plan, (uint32_t)row_bytes, (const uint32_t *)nullptr);
    }
    if (parts > 1) {
        ProfScope p("k_direct_fold", st);
        hipLaunchKernelGGL(k_direct_fold, dim3((unsigned)n_blobs), dim3(64), 0, st, partials, sums, parts);
    }
}

void launch_direct_msm(int bits, const uint64_t *table, size_t row_bytes, const uint32_t *scalars_raw, G1Xyzz29 *lane_scratch,
                       G1Xyzz29 *partials, uint32_t *redo, G1Xyzz29 *sums, size_t n_blobs, hipStream_t st, int fill, uint32_t *redo_flag_out) {
    const DirectPlanRt plan = make_plan(bits);
    if (!plan.entries) return;
    switch (bits) {
        case 14: launch_direct_t<14>(plan, table, row_bytes, scalars_raw, lane_scratch, partials, redo, sums, n_blobs, st, fill, redo_flag_out); break;
        case 15: launch_direct_t<15>(plan, table, row_bytes, scalars_raw, lane_scratch, partials, redo, sums, n_blobs, st, fill, redo_flag_out); break;
        case 16: launch_direct_t<16>(plan, table, row_bytes, scalars_raw, lane_scratch, partials, redo, sums, n_blobs, st, fill, redo_flag_out); break;
        default: launch_direct_t<0>(plan, table, row_bytes, scalars_raw, lane_scratch, partials, redo, sums, n_blobs, st, fill, redo_flag_out); break;  // 10 .. 13
    }
}
// words behind the redo flag that ONE blob's cooperative launch expects cleared (its hand-off counters), or 0 when a one-blob call on this
// table would not take the cooperative kernel (switched off, or a geometry that does not fit: the caller then keeps the plain path)
uint32_t direct_one_blob_counter_words(int bits) {
    const DirectPlanRt plan = make_plan(bits);
    if (!plan.entries || coop_max_blobs() < 1) return 0;
    CoopParams prm{};
    uint32_t units = 0, counters = 0;
    coop_geometry(plan, coop_rows_per_quad(plan.nw, 1), prm.n0, units, counters);
    if (!(units <= (uint32_t)kBlobElems && 1 + counters <= (uint32_t)kNumBuckets + 1)) return 0;
    return counters;
}
template <int CT>
static void launch_direct_only_t(const DirectPlanRt &plan, const uint64_t *table, size_t row_bytes, const uint32_t *scalars_raw, G1Xyzz29 *sums,
                                 const uint32_t *only_if, size_t n_blobs, hipStream_t st) {
    ProfScope p("k_direct_redo", st);
    hipLaunchKernelGGL(k_direct_accumulate<CT>, dim3(1, (unsigned)n_blobs, 1), dim3(kDirThreads), 0, st, table, (const uint4 *)scalars_raw, sums,
                       kBlobElems / kDirThreads, plan, (uint32_t)row_bytes, only_if);
}

void launch_direct_msm_only(int bits, const uint64_t *table, size_t row_bytes, const uint32_t *scalars_raw, G1Xyzz29 *sums, const uint32_t *only_if,
                            size_t n_blobs, hipStream_t st) {
    const DirectPlanRt plan = make_plan(bits);
    if (!plan.entries) return;
    switch (bits) {
        case 14: launch_direct_only_t<14>(plan, table, row_bytes, scalars_raw, sums, only_if, n_blobs, st); break;
        case 15: launch_direct_only_t<15>(plan, table, row_bytes, scalars_raw, sums, only_if, n_blobs, st); break;
        case 16: launch_direct_only_t<16>(plan, table, row_bytes, scalars_raw, sums, only_if, n_blobs, st); break;
        default: launch_direct_only_t<0>(plan, table, row_bytes, scalars_raw, sums, only_if, n_blobs, st); break;
    }
}

}  // namespace lwk
