// verify.hip -- verify_kzg_proof / verify_blob_kzg_proof / verify_blob_kzg_proof_batch
// (/root/reference/src/lib.rs:407-505, 525-692).
//
// Division of labour: everything per-blob -- point validation, the Fiat-Shamir
// challenge, the evaluation y = p(z), and the random linear combination of the batch -- runs on the
// GPU with the kernels the proof path already has; the two pairings per call run on the host
// (pairing.hip), as in the reference.
//
// The reference checks  e(C - [y]G, G2) * e(-pi, [tau]G2 - [z]G2) == 1  (KZG::verify). By bilinearity
// that is  e(C - [y]G + [z]pi, G2) * e(-pi, [tau]G2) == 1, which needs no G2 arithmetic: the same
// accept/reject bit with G2 and [tau]G2 taken straight from the setup.
#include <chrono>
#include "engine.h"
#include "knobs.h"
#include <memory>
#include "fp2.h"
#include "hostfp.h"
#include "glv.cuh"
#include <mutex>

#include <string.h>

#include <functional>
#include <thread>
#include <vector>

namespace lwk {

bool pairing_product_is_one(const G1Affine *ps, const Fp2 *qx, const Fp2 *qy, int n);
bool pairing_check_compressed(const uint8_t *g1s, const uint8_t *g2s, int n, bool *ok);
void sha256_host(uint8_t out[32], const uint8_t *msg, size_t len);
void sha256_fast(uint8_t out[32], const uint8_t *msg, size_t len);  // sha256_host.hip: SHA extensions when present
void sha256_fast_prefixed(uint8_t out[32], const uint8_t *prefix, size_t prefix_len, const uint8_t *msg, size_t len);
unsigned host_threads();  // sha256_host.hip: hardware threads capped by the cgroup quota
void host_parallel_for(size_t n, const std::function<void(size_t)> &fn);  // sha256_host.hip: persistent workers

namespace {

// reference blst_fp (canonical, most-significant u64 first) -> Fp
Fp fp_from_blst(const blst_fp &v) {
    uint32_t raw[12];
    for (int k = 0; k < 6; k++) {
        raw[2 * k] = (uint32_t)v.l[5 - k];
        raw[2 * k + 1] = (uint32_t)(v.l[5 - k] >> 32);
    }
    return fe_from_raw<FpParams>(raw);
}

// host-side G1 arithmetic runs on the 64-bit-limb field of hostfp.h through g1.cuh's generic formulas
typedef XyzzT<HFp, HFp, HFp> HXyzz;

struct HostPoint {
    G1Affine a;
    bool inf;
};

const HFp &host_beta() {
    static const HFp b = []() {
        uint32_t braw[12];
        g1_beta_raw(braw);
        return HFp::from_fe(fe_from_raw<FpParams>(braw));
    }();
    return b;
}

// a^((p + 1) / 4) on the 64-bit field, four bits of the exponent at a time (p = 3 mod 4: the square root when there is one)
HFp hfp_pow_quarter(const HFp &a) {
    static const uint32_t e[12] = {0xffffeaabu, 0xee7fbfffu, 0xac54ffffu, 0x07aaffffu, 0x3dac3d89u, 0xd9cc34a8u,
                                   0x3ce144afu, 0xd91dd2e1u, 0x90d2eb35u, 0x92c6e9edu, 0x8e5ff9a6u, 0x0680447au};
    HFp tab[16];
    tab[0] = HFp::one();
    for (int i = 1; i < 16; i++) tab[i] = tab[i - 1] * a;
    HFp acc = tab[(e[11] >> 28) & 15u];
    for (int w = 94; w >= 0; w--) {
        acc = sqr(sqr(sqr(sqr(acc))));
        const uint32_t d = (e[w >> 3] >> (4 * (w & 7))) & 15u;
        if (d) acc = acc * tab[d];
    }
    return acc;
}

// g1_decompress_nocheck (g1.cuh; /root/reference/src/compression.rs:62-103 without the subgroup check) with the square root on the host's
// own field: the 32-bit-limb form of g1.cuh exists for the GPU and its 381-bit power cost 0.1 ms of a one-blob verification here.
// Same return codes: 0 = affine point, 1 = infinity, 2 = invalid.
int host_decompress_nocheck(G1Affine &out, const uint8_t in[48]) {
    const uint8_t prefix = in[0] >> 5;
    if (!(prefix & 4)) return 2;
    if (prefix & 2) return 1;
    uint8_t b[48];
    memcpy(b, in, 48);
    b[0] &= 0x1f;
    uint32_t raw[12];
    raw_from_be<12>(raw, b);
    const Fp x32 = fe_from_raw<FpParams>(raw);   // x >= p is reduced, as in g1.cuh
    const HFp x = HFp::from_fe(x32);
    const HFp y2 = sqr(x) * x + HFp::from_fe(fp_from_u32(4));
    const HFp y = hfp_pow_quarter(y2);
    if (!(sqr(y) == y2)) return 2;
    const Fp y32 = y.to_fe(), yn32 = neg(y).to_fe();
    uint32_t ry[12], ryn[12];
    fe_to_raw<FpParams>(ry, y32);
    fe_to_raw<FpParams>(ryn, yn32);
    const bool y_greater = raw_geq<12>(ry, ryn);
    const bool want_greater = (prefix & 1) != 0;   // select_sqrt_value_from_third_bit: the greater root iff bit 5 is set
    out.x = x32;
    out.y = (want_greater == y_greater) ? y32 : yn32;
    return 0;
}

// decompress_g1_point incl. subgroup check (compression.rs:62-103), host side
bool host_g1_decompress(HostPoint &out, const uint8_t in[48]) {
    out.a.x = Fp::zero();
    out.a.y = Fp::zero();
    int rc = host_decompress_nocheck(out.a, in);
    if (rc == 2) return false;
    out.inf = rc == 1;
    if (out.inf) return true;
    return g1_in_subgroup_endo<HXyzz>(HFp::from_fe(out.a.x), HFp::from_fe(out.a.y), host_beta());
}

HXyzz to_xyzz(const HostPoint &p) {
    return p.inf ? HXyzz::infinity() : HXyzz::from_affine(HFp::from_fe(p.a.x), HFp::from_fe(p.a.y));
}

// requires !p.is_inf(); back to the representation the pairing entry point and the device share
G1Affine h_to_affine(const HXyzz &p) {
    HFp i = inv(p.zz * p.zzz);
    G1Affine r;
    r.x = (p.x * (i * p.zzz)).to_fe();
    r.y = (p.y * (i * p.zz)).to_fe();
    return r;
}

HXyzz xyzz_neg(const HXyzz &p) {
    HXyzz r = p;
    r.y = neg(p.y);
    return r;
}

// [k]P, left-to-right double-and-add: the definition (and the path of a k that is not below r)
HXyzz scalar_mul_plain(const HXyzz &p, const uint32_t k[8]) {
    HXyzz acc = HXyzz::infinity();
    for (int i = 255; i >= 0; i--) {
        acc = xyzz_dbl(acc);
        if ((k[i >> 5] >> (i & 31)) & 1) acc = xyzz_add(acc, p);
    }
    return acc;
}

// [k]P for P in G1 and k < r: the split of glv.cuh, k = lo + hi z^2 with [z^2]P = -phi(P) = (beta x, -y), both halves 128 bits, read
// four bits at a time against the sixteen multiples of P and their images -- 128 doublings and at most 64 additions where the loop
// above has 256 and ~128 (0.14 -> 0.06 ms under a one-blob verification). The caller's points passed the subgroup test.
HXyzz scalar_mul(const HXyzz &p, const uint32_t k[8]) {
    if (p.is_inf()) return p;
    if (raw_geq<8>(k, FrParams::MOD)) return scalar_mul_plain(p, k);
    uint32_t lo[4], hi[4];
    split_by_z2_barrett(lo, hi, k);
    HXyzz t1[16], t2[16];
    t1[1] = p;
    t1[2] = xyzz_dbl(p);
    for (int i = 3; i < 16; i++) t1[i] = xyzz_add(t1[i - 1], p);
    const HFp &beta = host_beta();
    for (int i = 1; i < 16; i++) {
        t2[i] = t1[i];
        if (t2[i].is_inf()) continue;   // (cannot happen for a point of order r; kept so that the table is right for any input)
        t2[i].x = beta * t1[i].x;
        t2[i].y = neg(t1[i].y);
    }
    HXyzz acc = HXyzz::infinity();
    for (int w = 31; w >= 0; w--) {
        for (int d = 0; d < 4; d++) acc = xyzz_dbl(acc);   // returns at once while acc is still empty
        const uint32_t a = (lo[w >> 3] >> (4 * (w & 7))) & 15u, b = (hi[w >> 3] >> (4 * (w & 7))) & 15u;
        if (a) acc = xyzz_add(acc, t1[a]);
        if (b) acc = xyzz_add(acc, t2[b]);
    }
    return acc;
}

// [k]G for the generator of the settings (srs.powers_main_group[0]): every verification multiplies the SAME point, so its multiples
// j 16^w G (w < 64, j = 1..15) are kept in affine form per distinct generator -- 64 mixed additions and no doubling per product
// (0.14 -> 0.02 ms). Built at the first verification on a generator (960 additions, one inversion: under a millisecond).
struct FixedBase {
    HFp gx, gy;
    std::vector<HFp> x, y;   // [w * 15 + (j - 1)]
};
std::shared_ptr<const FixedBase> fixed_base_of(const G1Affine &g) {
    static std::mutex mu;
    static std::shared_ptr<const FixedBase> last;
    const HFp gx = HFp::from_fe(g.x), gy = HFp::from_fe(g.y);
    {
        std::lock_guard<std::mutex> lk(mu);
        if (last && last->gx == gx && last->gy == gy) return last;
    }
    auto fb = std::make_shared<FixedBase>();
    fb->gx = gx;
    fb->gy = gy;
    const size_t n = 64 * 15;
    std::vector<HXyzz> m(n);
    HXyzz base = HXyzz::from_affine(gx, gy);
    for (int w = 0; w < 64; w++) {
        m[15 * w] = base;
        for (int j = 1; j < 15; j++) m[15 * w + j] = xyzz_add(m[15 * w + j - 1], base);
        base = xyzz_add(m[15 * w + 14], base);   // 16 x
    }
    // to affine with one inversion (Montgomery's trick over zz * zzz); an entry at infinity (a generator of tiny order: not a
    // setup anyone verifies against) leaves the table unusable and the caller on the generic product
    std::vector<HFp> d(n), pre(n);
    for (size_t i = 0; i < n; i++) {
        if (m[i].is_inf()) return nullptr;
        d[i] = m[i].zz * m[i].zzz;
        pre[i] = i ? pre[i - 1] * d[i] : d[i];
    }
    HFp iv = inv(pre[n - 1]);
    fb->x.resize(n);
    fb->y.resize(n);
    for (size_t i = n; i-- > 0;) {
        const HFp di = i ? iv * pre[i - 1] : iv;   // 1 / (zz zzz)
        if (i) iv = iv * d[i];
        fb->x[i] = m[i].x * (di * m[i].zzz);
        fb->y[i] = m[i].y * (di * m[i].zz);
    }
    std::lock_guard<std::mutex> lk(mu);
    last = fb;
    return fb;
}

HXyzz generator_mul(const G1Affine &g, const uint32_t k[8]) {
    std::shared_ptr<const FixedBase> fb = fixed_base_of(g);
    if (!fb) return scalar_mul_plain(HXyzz::from_affine(HFp::from_fe(g.x), HFp::from_fe(g.y)), k);   // (only on-curve is known of it)
    HXyzz acc = HXyzz::infinity();
    for (int w = 0; w < 64; w++) {
        const uint32_t j = (k[w >> 3] >> (4 * (w & 7))) & 15u;
        if (j) acc = xyzz_madd(acc, fb->x[15 * w + j - 1], fb->y[15 * w + j - 1]);
    }
    return acc;
}

// field element bytes -> canonical limbs. reference mode: big-endian, reduced; c-kzg mode: little-endian, canonical
bool fr_from_bytes(uint32_t raw[8], const uint8_t b[32], int mode) {
    uint32_t t[8];
    if (mode == LWKZG_MODE_CKZG) {
        raw_from_le<8>(t, b);
        if (raw_geq<8>(t, FrParams::MOD)) return false;
    } else {
        raw_from_be<8>(t, b);
    }
    Fr f = fe_from_raw<FrParams>(t);
    fe_to_raw<FrParams>(raw, f);
    return true;
}

// e(lhs, G2) * e(-rhs_point, [tau]G2) == 1 with the G2 points of the settings
C_KZG_RET pairing_verdict(bool *ok, const HXyzz &lhs, const HXyzz &pi, const KZGSettings *s) {
    if (!s->g2_values) {
        set_error("KZGSettings.g2_values is NULL");
        return C_KZG_ERROR;
    }
    G1Affine ps[2];
    Fp2 qx[2], qy[2];
    int m = 0;
    const g2_t *g2 = s->g2_values;
    if (!lhs.is_inf()) {
        ps[m] = h_to_affine(lhs);
        qx[m] = {fp_from_blst(g2[0].x.fp[0]), fp_from_blst(g2[0].x.fp[1])};
        qy[m] = {fp_from_blst(g2[0].y.fp[0]), fp_from_blst(g2[0].y.fp[1])};
        m++;
    }
    if (!pi.is_inf()) {
        ps[m] = h_to_affine(xyzz_neg(pi));
        qx[m] = {fp_from_blst(g2[1].x.fp[0]), fp_from_blst(g2[1].x.fp[1])};
        qy[m] = {fp_from_blst(g2[1].y.fp[0]), fp_from_blst(g2[1].y.fp[1])};
        m++;
    }
    *ok = pairing_product_is_one(ps, qx, qy, m);
    return C_KZG_OK;
}

// generator as the reference takes it: srs.powers_main_group[0]
bool setup_generator(HostPoint &g, const KZGSettings *s) {
    if (!s->g1_values) return false;
    g.a.x = fp_from_blst(s->g1_values[0].x);
    g.a.y = fp_from_blst(s->g1_values[0].y);
    g.inf = false;
    return g1_on_curve(g.a);
}

C_KZG_RET verify_core(bool *ok, const HostPoint &c, const uint32_t z[8], const uint32_t y[8], const HostPoint &pi,
                      const KZGSettings *s) {
    HostPoint g;
    if (!setup_generator(g, s)) {
        set_error("g1_values[0] is not a curve point");
        return C_KZG_ERROR;
    }
    const bool timing = knobs().timing;
    const auto t0 = std::chrono::steady_clock::now();
    HXyzz zpi;
    SideTask side([&]() { zpi = scalar_mul(to_xyzz(pi), z); });  // the two scalar multiplications side by side
    HXyzz lhs = xyzz_add(to_xyzz(c), xyzz_neg(generator_mul(g.a, y)));  // C - [y]G
    const auto t1 = std::chrono::steady_clock::now();
    side.join();
    lhs = xyzz_add(lhs, zpi);                                               //   + [z]pi
    if (timing) {
        const auto t2 = std::chrono::steady_clock::now();
        fprintf(stderr, "[lambdaworks_kzg_amd] verification: C - [y]G %.3f ms, [z]pi beside it done after %.3f ms\n",
                std::chrono::duration<double, std::milli>(t1 - t0).count(), std::chrono::duration<double, std::milli>(t2 - t0).count());
    }
    return pairing_verdict(ok, lhs, to_xyzz(pi), s);
}

// The three linear combinations of a small batch on the host threads (g1_lincomb, lib.rs:679-685):
//   out[0] = sum r_i pi_i,  out[1] = sum (r_i z_i) pi_i,  out[2] = sum r_i C_i
// over the points the host validation kept (aff/kind: n commitments, then n proofs). The 3n terms are dealt out in
// contiguous runs, so a thread mostly works on one sum; within a run the terms share their doublings (Straus).
void host_lincomb3(HXyzz out[3], const G1Affine29 *aff, const int32_t *kind, const uint8_t *sc_r, const uint8_t *sc_rz, size_t n) {
    struct Term {
        HFp x, y;
        uint32_t k[8];
        int sum;
    };
    std::vector<Term> terms;
    terms.reserve(3 * n);
    for (int s = 0; s < 3; s++)
        for (size_t i = 0; i < n; i++) {
            const size_t p = s == 2 ? i : n + i;  // commitments first, then proofs
            if (kind[p] != 0) continue;           // the point at infinity adds nothing
            Term t;
            t.x = HFp::from_fe(f29_to_fp(aff[p].x));
            t.y = HFp::from_fe(f29_to_fp(aff[p].y));
            raw_from_be<8>(t.k, (s == 1 ? sc_rz : sc_r) + 32 * i);
            t.sum = s;
            terms.push_back(t);
        }
    for (int s = 0; s < 3; s++) out[s] = HXyzz::infinity();
    const size_t nterms = terms.size();
    if (nterms == 0) return;
    unsigned nt = host_threads();
    if (nt > nterms) nt = (unsigned)nterms;
    std::vector<HXyzz> part(3 * (size_t)nt, HXyzz::infinity());
    auto run = [&](size_t t) {
        const size_t lo = nterms * t / nt, hi = nterms * (t + 1) / nt;
        HXyzz acc[3] = {HXyzz::infinity(), HXyzz::infinity(), HXyzz::infinity()};
        for (int bit = 255; bit >= 0; bit--) {
            for (int s = 0; s < 3; s++) acc[s] = xyzz_dbl(acc[s]);  // returns at once while a sum is still empty
            for (size_t j = lo; j < hi; j++)
                if ((terms[j].k[bit >> 5] >> (bit & 31)) & 1) acc[terms[j].sum] = xyzz_madd(acc[terms[j].sum], terms[j].x, terms[j].y);
        }
        for (int s = 0; s < 3; s++) part[3 * t + s] = acc[s];
    };
    host_parallel_for(nt, run);
    for (unsigned t = 0; t < nt; t++)
        for (int s = 0; s < 3; s++) out[s] = xyzz_add(out[s], part[3 * t + s]);
}

C_KZG_RET bad(int mode) { return mode == LWKZG_MODE_CKZG ? C_KZG_BADARGS : C_KZG_ERROR; }

}  // namespace
}  // namespace lwk

namespace lwk {
int host_validate_commitment(const uint8_t in48[48], uint8_t canon48[48], G1Affine29 *aff) {
    HostPoint p;
    memset(canon48, 0, 48);
    if (aff) {
        aff->x = F29<2>::zero();
        aff->y = F29<2>::zero();
    }
    if (!host_g1_decompress(p, in48)) return 2;
    if (p.inf) {
        canon48[0] = 0xc0;
        return 1;
    }
    g1_compress_affine(canon48, p.a);
    if (aff) *aff = affine_to_29(p.a);  // what k_validate_commitments leaves on the device for the linear combinations
    return 0;
}

// the same for n points, spread over the host threads (~0.2 ms per point per thread)
void host_validate_commitments(const uint8_t *in48, uint8_t *canon48, int *rc, size_t n, G1Affine29 *aff) {
    host_parallel_for(n, [=](size_t i) { rc[i] = host_validate_commitment(in48 + 48 * i, canon48 + 48 * i, aff ? aff + i : nullptr); });
}
}  // namespace lwk

using namespace lwk;

extern "C" {

C_KZG_RET verify_kzg_proof(bool *ok, const Bytes48 *commitment_bytes, const Bytes32 *z_bytes, const Bytes32 *y_bytes,
                           const Bytes48 *proof_bytes, const KZGSettings *s) {
    if (!ok) return C_KZG_BADARGS;
    *ok = false;  // lib.rs:415-417
    const int mode = mode_of(s);
    if (!commitment_bytes || !z_bytes || !y_bytes || !proof_bytes || !s) return bad(mode);
    HostPoint c, pi;
    uint32_t z[8], y[8];
    // order of the reference: commitment, z, y, proof (lib.rs:424-440)
    // (every failure here is the same return code, so the two decompressions may run side by side)
    bool pi_ok = false;
    const auto t0 = std::chrono::steady_clock::now();
    SideTask side([&]() { pi_ok = host_g1_decompress(pi, proof_bytes->bytes); });
    const bool c_ok = host_g1_decompress(c, commitment_bytes->bytes);
    const auto t1 = std::chrono::steady_clock::now();
    side.join();
    if (knobs().timing)
        fprintf(stderr, "[lambdaworks_kzg_amd] verification: commitment decompressed + subgroup test %.3f ms, the proof beside it done after %.3f ms\n",
                std::chrono::duration<double, std::milli>(t1 - t0).count(),
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    if (!c_ok) { set_error("invalid commitment"); return bad(mode); }
    if (!fr_from_bytes(z, z_bytes->bytes, mode)) { set_error("z is not canonical"); return bad(mode); }
    if (!fr_from_bytes(y, y_bytes->bytes, mode)) { set_error("y is not canonical"); return bad(mode); }
    if (!pi_ok) { set_error("invalid proof"); return bad(mode); }
    return verify_core(ok, c, z, y, pi, s);
}

C_KZG_RET verify_blob_kzg_proof(bool *ok, const Blob *blob, const Bytes48 *commitment_bytes, const Bytes48 *proof_bytes,
                                const KZGSettings *s) {
    if (!ok) return C_KZG_BADARGS;
    *ok = false;  // lib.rs:463-465
    const int mode = mode_of(s);
    if (!blob || !commitment_bytes || !proof_bytes || !s) return bad(mode);
    HostPoint c, pi;
    // lib.rs:473-478: both points are decompressed before the blob is parsed (every failure of this function has the
    // same return code in either mode, so the order is not observable). Here both decompressions (0.2 ms each: square
    // root + subgroup test) run on threads of their own BESIDE the per-blob GPU pass, which is started on the
    // assumption that the caller's commitment bytes are the canonical encoding -- what the challenge hashes
    // (utils.rs:138) -- and is repeated with the canonical bytes in the rare case that they are not (an infinity
    // encoding with stray bits). Decompressing here also validates: the GPU pass needs no validation kernel.
    bool pi_ok = false, c_ok = false;
    SideTask side_pi([&]() { pi_ok = host_g1_decompress(pi, proof_bytes->bytes); });  // joined by their destructors on
    SideTask side_c([&]() { c_ok = host_g1_decompress(c, commitment_bytes->bytes); });  // every exit
    Ctx *ctx = ctx_of(s);
    if (!ctx) return C_KZG_ERROR;
    uint8_t zb[32], yb[32], canon[48], canon_in[48];
    C_KZG_RET rc;
    {
        VerifyBuffers vb;
        rc = verify_prepare_host(ctx, blob->bytes, commitment_bytes->bytes, nullptr, 1, mode, zb, yb, canon, nullptr, vb,
                                 commitment_bytes->bytes);
    }
    side_c.join();
    if (!c_ok) { set_error("invalid commitment"); return bad(mode); }
    if (c.inf) {
        memset(canon_in, 0, 48);
        canon_in[0] = 0xc0;
    } else {
        g1_compress_affine(canon_in, c.a);
    }
    if (memcmp(canon_in, commitment_bytes->bytes, 48) != 0) {  // valid, but not the canonical bytes: hash those instead
        VerifyBuffers vb;
        rc = verify_prepare_host(ctx, blob->bytes, commitment_bytes->bytes, nullptr, 1, mode, zb, yb, canon, nullptr, vb, canon_in);
    }
    side_pi.join();
    if (!pi_ok) { set_error("invalid proof"); return bad(mode); }
    if (rc != C_KZG_OK) return mode == LWKZG_MODE_REFERENCE ? C_KZG_ERROR : rc;
    uint32_t z[8], y[8];
    if (!fr_from_bytes(z, zb, mode) || !fr_from_bytes(y, yb, mode)) return C_KZG_ERROR;
    return verify_core(ok, c, z, y, pi, s);
}

// ---- batch verification, in the three steps a sharded run needs -----------------------------------------------------
// verify_kzg_proof_batch, lib.rs:639-692: r from SHA-256 over
//   "RCKZGBATCH___V1_" | usize(4096) LE | usize(n) LE | n x (C 48 | z 32 | y 32 | pi 48)      (utils.rs:166-206)
// powers 1, r, r^2, ...;  rhs = sum r^i (C_i - [y_i]G) + sum r^i z_i pi_i ;  e(rhs, G2) == e(sum r^i pi_i, [tau]G2).
//
//   begin   : everything per blob (validate C_i, pi_i; z_i = challenge; y_i = p_i(z_i)) -> the blob's 160-byte record
//             of the transcript; the decompressed points stay on the device
//   partial : r from the WHOLE transcript; this shard's terms of the three linear combinations and of sum r^i y_i
//   finish  : add the shards' partial sums, one pairing check
// One process: begin, partial(first = 0, n_total = n), finish. Eight GPUs: every rank begins its shard, the records are
// all-gathered (160 bytes per blob), every rank computes its partial sums with the common r, the partial sums (328
// bytes per rank) are all-gathered and every rank finishes: SURVEY section 8e, the reference's computation exactly.
}  // extern "C"

namespace lwk {
namespace {

constexpr size_t kRecord = LWKZG_VERIFY_RECORD_BYTES;    // C 48 | z 32 | y 32 | pi 48
constexpr size_t kPartial = LWKZG_VERIFY_PARTIAL_BYTES;  // 3 x (flag 1 | x 48 | y 48) | sum r^i y_i 32 | pad

struct Shard {
    Ctx *ctx = nullptr;
    uint64_t ctx_generation = 0;   // the context's identity (engine.h: ctx_is_live): the caller's KZGSettings is never read again
    const KZGSettings *s = nullptr;
    int device = 0;   // kept here: the shard may outlive its settings' context (a caller that frees the setup first)
    size_t n = 0;
    int mode = 0;
    VerifyBuffers vb;
    std::vector<uint8_t> records;   // C | z | y | pi per blob (160 n): the shard's part of the transcript, and where z_i / y_i are read from
    const uint8_t *z(size_t i) const { return &records[kRecord * i + 48]; }
    const uint8_t *y(size_t i) const { return &records[kRecord * i + 80]; }
    ~Shard() {
        if (vb.owned) {  // device memory of the shard's own: freed on its device whether or not the context still exists
            hipSetDevice(device);
            hipDeviceSynchronize();
            verify_buffers_free(vb);
        }
    }
};

// device_inputs: blobs / comms / proofs are DEVICE pointers, produced on `caller` (may be null)
C_KZG_RET shard_begin(Shard &sh, const uint8_t *blobs, const uint8_t *comms, const uint8_t *proofs, size_t n, const KZGSettings *s,
                      int mode, bool own_buffers, bool device_inputs = false, hipStream_t caller = nullptr) {
    sh.ctx = ctx_of(s);
    if (!sh.ctx) return C_KZG_ERROR;
    sh.device = sh.ctx->device;
    sh.ctx_generation = sh.ctx->generation;
    sh.s = s;
    sh.n = n;
    sh.mode = mode;
    sh.vb.owned = own_buffers;
    sh.records.resize(kRecord * n);
    if (n == 0) return C_KZG_OK;
    // per blob on the GPU: validate C_i and pi_i (decompress + subgroup check + canonical recompression; the
    // decompressed points stay on the device), z_i = challenge(blob_i, C_i), y_i = p_i(z_i)
    C_KZG_RET rc;
    if (device_inputs) {   // the transcript is assembled on the device and arrives in one copy (engine.hip)
        rc = verify_prepare_device(sh.ctx, blobs, comms, proofs, n, mode, nullptr, nullptr, nullptr, nullptr, sh.vb, caller, sh.records.data());
    } else {
        std::vector<uint8_t> zs(32 * n), ys(32 * n), canon_c(48 * n), canon_p(48 * n);
        rc = verify_prepare_host(sh.ctx, blobs, comms, proofs, n, mode, zs.data(), ys.data(), canon_c.data(), canon_p.data(), sh.vb);
        if (rc == C_KZG_OK)
            for (size_t i = 0; i < n; i++) {   // z and y enter in the mode's byte order, as to_bytes_be / c-kzg's bytes_from_bls_field do
                uint8_t *m = &sh.records[kRecord * i];
                memcpy(m, &canon_c[48 * i], 48);
                memcpy(m + 48, &zs[32 * i], 32);
                memcpy(m + 80, &ys[32 * i], 32);
                memcpy(m + 112, &canon_p[48 * i], 48);
            }
    }
    if (rc != C_KZG_OK) return mode == LWKZG_MODE_REFERENCE ? C_KZG_ERROR : rc;
    return C_KZG_OK;
}

// this shard's records
void shard_records(const Shard &sh, uint8_t *out) {
    if (sh.n) memcpy(out, sh.records.data(), kRecord * sh.n);
}

HFr hfr_raw(const uint32_t t[8]) {
    HFr x;
    for (int k = 0; k < 4; k++) x.l[k] = (uint64_t)t[2 * k] | ((uint64_t)t[2 * k + 1] << 32);
    return x;
}
void hfr_to_raw(uint32_t t[8], const HFr &x) {
    for (int k = 0; k < 4; k++) {
        t[2 * k] = (uint32_t)x.l[k];
        t[2 * k + 1] = (uint32_t)(x.l[k] >> 32);
    }
}
void hfr_to_be(uint8_t *out, const HFr &x) {
    uint32_t t[8];
    hfr_to_raw(t, x);
    raw_to_be<8>(out, t);
}

// r (utils.rs:166-206) from the n_total records of the whole batch, in Montgomery form
HFr batch_challenge_mont(const uint8_t *records, size_t n_total, bool le) {
    uint8_t head[32];
    memcpy(head, "RCKZGBATCH___V1_", 16);
    memset(head + 16, 0, 16);
    head[16] = 0x00;
    head[17] = 0x10;  // 4096 LE
    for (int k = 0; k < 8; k++) head[24 + k] = (uint8_t)((uint64_t)n_total >> (8 * k));
    uint8_t dg[32];
    sha256_fast_prefixed(dg, head, 32, records, n_total * kRecord);   // (no 160 n-byte copy of the transcript behind its header)
    uint32_t t[8], rraw[8];
    if (le) raw_from_le<8>(t, dg); else raw_from_be<8>(t, dg);
    Fr f = fe_from_raw<FrParams>(t);  // hash_field_unsafe: reduced mod r
    fe_to_raw<FrParams>(rraw, f);
    return hfr_raw(rraw) * hfr_raw(FrParams::R2);
}

// The shard's terms i = first .. first + n - 1 of the batch: out[0] = sum r^i pi_i, out[1] = sum r^i z_i pi_i,
// out[2] = sum r^i C_i (g1_lincomb, lib.rs:679-685), ysum = sum r^i y_i (raw). `beside(ysum)` (optional) is called once
// the scalars are known and before the linear combinations run (the single-process path starts [ysum]G there).
C_KZG_RET shard_partial(Shard &sh, const uint8_t *records_all, size_t n_total, size_t first, HXyzz out[3], HFr &ysum,
                        const std::function<void(const HFr &)> &beside) {
    const bool le = sh.mode == LWKZG_MODE_CKZG;
    const size_t n = sh.n;
    for (int k = 0; k < 3; k++) out[k] = HXyzz::infinity();
    ysum = HFr::zero();
    if (first + n > n_total) {
        set_error("verify shard [%zu, %zu) does not fit a batch of %zu", first, first + n, n_total);
        return C_KZG_BADARGS;
    }
    if (n && memcmp(records_all + kRecord * first + 48, sh.z(0), 32) != 0) {
        set_error("verify shard: the transcript at index %zu is not this shard's first record", first);
        return C_KZG_BADARGS;
    }
    const HFr r_mont = batch_challenge_mont(records_all, n_total, le);
    // scalars r^i, r^i z_i (for the GPU) and sum r^i y_i (one host scalar), on the 64-bit host field. The running power
    // stays in Montgomery form; a Montgomery product with a RAW factor (z_i, y_i, 1) yields the raw product directly.
    const uint32_t one_limbs[8] = {1, 0, 0, 0, 0, 0, 0, 0};
    const HFr one_raw = hfr_raw(one_limbs);
    HFr rp = HFr::one();  // r^first, square and multiply
    {
        HFr base = r_mont;
        for (size_t e = first; e; e >>= 1) {
            if (e & 1) rp = rp * base;
            base = base * base;
        }
    }
    // three variable-base linear combinations: on the GPU over the points the validation kernels left there, or on the
    // host threads when the batch was small enough to be validated there
    const bool on_host = sh.vb.h_aff.size() == 2 * n && sh.vb.h_kind.size() == 2 * n;
    // r06: the GPU derives its scalars from r itself (vmsm.hip: r^(2^k) and r^first go up, 1 KiB) and is at work on the sums while this
    // thread walks the powers for sum r^i y_i; r05's arm needs the 2 x 32 n scalar bytes from here first
    const bool dev_msm = n != 0 && !on_host && vmsm_ready(sh.vb);
    if (dev_msm) {
        Fr pw[33];
        HFr sq = r_mont;
        for (int k = 0; k < 32; k++) {
            pw[k] = sq.to_fe();
            sq = sq * sq;
        }
        pw[32] = rp.to_fe();
        if (vmsm_begin(sh.ctx, sh.vb, pw, le ? 1 : 0, n) != C_KZG_OK) return C_KZG_ERROR;
    }
    std::vector<uint8_t> sc_r(dev_msm ? 0 : 32 * n), sc_rz(dev_msm ? 0 : 32 * n);
    for (size_t i = 0; i < n; i++) {
        uint32_t zr[8], yr[8];
        if (le) {
            raw_from_le<8>(zr, sh.z(i));
            raw_from_le<8>(yr, sh.y(i));
        } else {
            raw_from_be<8>(zr, sh.z(i));
            raw_from_be<8>(yr, sh.y(i));
        }
        if (raw_geq<8>(zr, FrParams::MOD) || raw_geq<8>(yr, FrParams::MOD)) return C_KZG_ERROR;  // the GPU wrote canonical values
        if (!dev_msm) {
            hfr_to_be(&sc_r[32 * i], rp * one_raw);
            hfr_to_be(&sc_rz[32 * i], rp * hfr_raw(zr));
        }
        ysum = ysum + rp * hfr_raw(yr);
        rp = rp * r_mont;
    }
    if (beside) beside(ysum);
    if (n == 0) return C_KZG_OK;
    if (on_host) {
        host_lincomb3(out, sh.vb.h_aff.data(), sh.vb.h_kind.data(), sc_r.data(), sc_rz.data(), n);
        return C_KZG_OK;
    }
    uint8_t sums[3][96];
    int infs[3];
    C_KZG_RET rc = dev_msm ? vmsm_finish(sh.ctx, sh.vb, sums, infs) : lincomb3_device_host(sh.ctx, sh.vb, sc_r.data(), sc_rz.data(), n, sums, infs);
    if (rc != C_KZG_OK) return C_KZG_ERROR;
    for (int k = 0; k < 3; k++) {
        if (infs[k]) continue;
        uint32_t raw[12];
        G1Affine a;
        raw_from_be<12>(raw, sums[k]);
        a.x = fe_from_raw<FpParams>(raw);
        raw_from_be<12>(raw, sums[k] + 48);
        a.y = fe_from_raw<FpParams>(raw);
        out[k] = HXyzz::from_affine(HFp::from_fe(a.x), HFp::from_fe(a.y));
    }
    return C_KZG_OK;
}

// e(sum r^i C_i - [sum r^i y_i]G + sum r^i z_i pi_i, G2) == e(sum r^i pi_i, [tau]G2): kzg.verify(0, 0, rhs, proof_lincomb), lib.rs:691
C_KZG_RET batch_verdict(bool *ok, const HXyzz sums[3], const HXyzz &ysum_g, const KZGSettings *s) {
    HXyzz rhs = xyzz_add(sums[2], xyzz_neg(ysum_g));
    rhs = xyzz_add(rhs, sums[1]);
    return pairing_verdict(ok, rhs, sums[0], s);
}

void point_to_bytes(uint8_t *out97, const HXyzz &p) {
    memset(out97, 0, 97);
    if (p.is_inf()) {
        out97[0] = 1;
        return;
    }
    const G1Affine a = h_to_affine(p);
    uint32_t raw[12];
    fe_to_raw<FpParams>(raw, a.x);
    raw_to_be<12>(out97 + 1, raw);
    fe_to_raw<FpParams>(raw, a.y);
    raw_to_be<12>(out97 + 49, raw);
}

bool point_from_bytes(HXyzz &p, const uint8_t *in97) {
    if (in97[0] == 1) {
        p = HXyzz::infinity();
        return true;
    }
    if (in97[0] != 0) return false;
    uint32_t raw[12];
    G1Affine a;
    raw_from_be<12>(raw, in97 + 1);
    if (raw_geq<12>(raw, FpParams::MOD)) return false;
    a.x = fe_from_raw<FpParams>(raw);
    raw_from_be<12>(raw, in97 + 49);
    if (raw_geq<12>(raw, FpParams::MOD)) return false;
    a.y = fe_from_raw<FpParams>(raw);
    if (!g1_on_curve(a)) return false;
    p = HXyzz::from_affine(HFp::from_fe(a.x), HFp::from_fe(a.y));
    return true;
}

}  // namespace
}  // namespace lwk

extern "C" {

}  // extern "C"
namespace {
// the verification entry points size host vectors by n: nothing may unwind across the C ABI
template <class F>
C_KZG_RET guarded(const char *what, F &&f) {
    try {
        return f();
    } catch (const std::bad_alloc &) {
        lwk::set_error("%s: out of host memory", what);
        return C_KZG_MALLOC;
    } catch (...) {
        lwk::set_error("%s: unexpected exception", what);
        return C_KZG_ERROR;
    }
}
}  // namespace
extern "C" {

static C_KZG_RET verify_batch_impl(bool *ok, const Blob *blobs, const Bytes48 *commitments_bytes, const Bytes48 *proofs_bytes, size_t n,
                                   const KZGSettings *s, bool device_inputs = false, hipStream_t caller = nullptr) {
    if (!ok) return C_KZG_BADARGS;
    *ok = false;  // lib.rs:533-535
    const int mode = mode_of(s);
    if (n == 0) {
        // the reference answers OK with ok = false (lib.rs:538-543); c-kzg-4844 accepts the empty batch
        // (verify_blob_kzg_proof_batch_case_a271b78b8e869d69: output true) -- SURVEY Appendix B: a reference quirk, never in mode C
        *ok = mode == LWKZG_MODE_CKZG;
        return C_KZG_OK;
    }
    if (n == 1 && !device_inputs) return verify_blob_kzg_proof(ok, blobs, commitments_bytes, proofs_bytes, s);  // lib.rs:544
    if (!blobs || !commitments_bytes || !proofs_bytes || !s) return bad(mode);

    const bool timing = knobs().timing;  // phase wall-clock to stderr
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
        return std::chrono::duration<double, std::milli>(b - a).count();
    };
    const auto t0 = now();
    Shard sh;
    C_KZG_RET rc = shard_begin(sh, (const uint8_t *)blobs, (const uint8_t *)commitments_bytes, (const uint8_t *)proofs_bytes, n, s,
                               mode, false, device_inputs, caller);
    if (rc != C_KZG_OK) return rc;
    const auto t1 = now();
    const std::vector<uint8_t> &records = sh.records;   // one shard: its records are the transcript
    HostPoint g;
    if (!setup_generator(g, s)) return C_KZG_ERROR;
    // [sum r^i y_i]G on a thread of its own, beside the linear combinations
    HXyzz ysum_g = HXyzz::infinity();
    SideTask side;
    auto t2 = t1;
    HXyzz sums[3];
    HFr ysum;
    rc = shard_partial(sh, records.data(), n, 0, sums, ysum, [&](const HFr &ys) {
        t2 = now();
        side.start([&, ys]() {
            uint32_t ys_raw[8];
            hfr_to_raw(ys_raw, ys);
            ysum_g = generator_mul(g.a, ys_raw);
        });
    });
    side.join();
    if (rc != C_KZG_OK) return C_KZG_ERROR;
    const auto t3 = now();
    rc = batch_verdict(ok, sums, ysum_g, s);
    if (timing)
        fprintf(stderr, "[lambdaworks_kzg_amd] verify batch n=%zu: prepare (H2D, validate, challenge, evaluate) %.2f ms, "
                        "r powers %.2f ms, lincomb3 %.2f ms, pairing side %.2f ms\n",
                n, ms(t0, t1), ms(t1, t2), ms(t2, t3), ms(t3, now()));
    return rc;
}

C_KZG_RET verify_blob_kzg_proof_batch(bool *ok, const Blob *blobs, const Bytes48 *commitments_bytes,
                                      const Bytes48 *proofs_bytes, size_t n, const KZGSettings *s) {
    if (ok) *ok = false;
    return guarded("verify_blob_kzg_proof_batch", [&] { return verify_batch_impl(ok, blobs, commitments_bytes, proofs_bytes, n, s); });
}

// verify_blob_kzg_proof_batch for a batch that is already in HBM (a producer that committed and proved on the GPU, a node that received
// its blobs by RDMA): device pointers in, the verdict out. Synchronous like the reference's call -- the verdict is a host bool and the
// pairing runs on the host -- but nothing crosses PCIe except the 160-byte records (20.5 ms -> the kernels' ~7 ms at 4096 blobs).
C_KZG_RET lwkzg_verify_blob_kzg_proof_batch_device(bool *ok, const void *blobs_dev, const void *commitments48_dev, const void *proofs48_dev,
                                                   size_t n, const KZGSettings *s, void *stream) {
    if (ok) *ok = false;
    return guarded("lwkzg_verify_blob_kzg_proof_batch_device", [&] {
        return verify_batch_impl(ok, (const Blob *)blobs_dev, (const Bytes48 *)commitments48_dev, (const Bytes48 *)proofs48_dev, n, s, true,
                                 (hipStream_t)stream);
    });
}

// ---- the same three steps for a batch sharded over several processes / GPUs (include/lambdaworks_kzg_amd.h) ----------

static C_KZG_RET shard_begin_impl(LwkzgVerifyShard **shard_out, uint8_t *records_out, const Blob *blobs, const Bytes48 *commitments,
                                  const Bytes48 *proofs, size_t n_local, const KZGSettings *s, bool device_inputs = false,
                                  hipStream_t caller = nullptr) {
    if (!shard_out) return C_KZG_BADARGS;
    *shard_out = nullptr;
    const int mode = mode_of(s);
    if (!s || (n_local && (!records_out || !blobs || !commitments || !proofs))) return bad(mode);
    std::unique_ptr<Shard> sh(new (std::nothrow) Shard());   // (owned until handed out: shard_begin's host vectors may throw)
    if (!sh) return C_KZG_MALLOC;
    C_KZG_RET rc = shard_begin(*sh, (const uint8_t *)blobs, (const uint8_t *)commitments, (const uint8_t *)proofs, n_local, s, mode, true,
                               device_inputs, caller);
    if (rc != C_KZG_OK) return rc;
    shard_records(*sh, records_out);
    *shard_out = (LwkzgVerifyShard *)sh.release();
    return C_KZG_OK;
}

C_KZG_RET lwkzg_verify_shard_begin(LwkzgVerifyShard **shard_out, uint8_t *records_out, const Blob *blobs, const Bytes48 *commitments,
                                   const Bytes48 *proofs, size_t n_local, const KZGSettings *s) {
    return guarded("lwkzg_verify_shard_begin", [&] { return shard_begin_impl(shard_out, records_out, blobs, commitments, proofs, n_local, s); });
}

// the shard's blobs, commitments and proofs as DEVICE pointers (produced on `stream`, which may be null); the records come back to the host
C_KZG_RET lwkzg_verify_shard_begin_device(LwkzgVerifyShard **shard_out, uint8_t *records_out, const void *blobs_dev, const void *commitments48_dev,
                                          const void *proofs48_dev, size_t n_local, const KZGSettings *s, void *stream) {
    return guarded("lwkzg_verify_shard_begin_device", [&] {
        return shard_begin_impl(shard_out, records_out, (const Blob *)blobs_dev, (const Bytes48 *)commitments48_dev, (const Bytes48 *)proofs48_dev,
                                n_local, s, true, (hipStream_t)stream);
    });
}

static C_KZG_RET shard_partial_impl(uint8_t *partial_out, LwkzgVerifyShard *shard, const uint8_t *records_all, size_t n_total,
                                    size_t first_index) {
    if (!partial_out || !shard || (!records_all && n_total)) return C_KZG_BADARGS;
    Shard &sh = *(Shard *)shard;
    if (!ctx_is_live(sh.ctx, sh.ctx_generation)) {  // the setup was freed (or rebuilt) under the shard: its context is gone (decided from the registry,
                                                     // not by dereferencing the caller's settings pointer, which may be freed or reused by now)
        set_error("lwkzg_verify_shard_partial: the shard's trusted setup is no longer loaded");
        return C_KZG_BADARGS;
    }
    HXyzz sums[3];
    HFr ysum;
    C_KZG_RET rc = shard_partial(sh, records_all, n_total, first_index, sums, ysum, nullptr);
    if (rc != C_KZG_OK) return rc;
    memset(partial_out, 0, kPartial);
    for (int k = 0; k < 3; k++) point_to_bytes(partial_out + 97 * k, sums[k]);
    hfr_to_be(partial_out + 291, ysum);
    return C_KZG_OK;
}

C_KZG_RET lwkzg_verify_shard_partial(uint8_t *partial_out, LwkzgVerifyShard *shard, const uint8_t *records_all, size_t n_total,
                                     size_t first_index) {
    return guarded("lwkzg_verify_shard_partial", [&] { return shard_partial_impl(partial_out, shard, records_all, n_total, first_index); });
}

void lwkzg_verify_shard_free(LwkzgVerifyShard *shard) { delete (Shard *)shard; }

C_KZG_RET lwkzg_verify_shards_finish(bool *ok, const uint8_t *partials, size_t n_shards, size_t n_total, const KZGSettings *s) {
    if (!ok) return C_KZG_BADARGS;
    *ok = false;
    if (n_total == 0) {  // lib.rs:538-543: the empty batch is OK with ok = false; c-kzg-4844 (mode C) accepts it
        *ok = mode_of(s) == LWKZG_MODE_CKZG;
        return C_KZG_OK;
    }
    if (!partials || !n_shards || !s) return C_KZG_BADARGS;
    HXyzz sums[3] = {HXyzz::infinity(), HXyzz::infinity(), HXyzz::infinity()};
    HFr ysum = HFr::zero();
    for (size_t r = 0; r < n_shards; r++) {
        const uint8_t *p = partials + kPartial * r;
        for (int k = 0; k < 3; k++) {
            HXyzz q;
            if (!point_from_bytes(q, p + 97 * k)) {
                set_error("lwkzg_verify_shards_finish: partial sum %d of shard %zu is not a curve point", k, r);
                return C_KZG_BADARGS;
            }
            sums[k] = xyzz_add(sums[k], q);
        }
        uint32_t t[8];
        raw_from_be<8>(t, p + 291);
        if (raw_geq<8>(t, FrParams::MOD)) return C_KZG_BADARGS;
        ysum = ysum + hfr_raw(t);
    }
    HostPoint g;
    if (!setup_generator(g, s)) return C_KZG_ERROR;
    uint32_t ys_raw[8];
    hfr_to_raw(ys_raw, ysum);
    return batch_verdict(ok, sums, generator_mul(g.a, ys_raw), s);
}

// sum of compressed points on the host (gathering the per-GPU partial sums of a sharded long MSM: SURVEY 8e,
// "one gather of 8 points + 7 host additions")
C_KZG_RET lwkzg_g1_sum_compressed(uint8_t out48[48], const uint8_t *points48, size_t n) {
    if (!out48 || (!points48 && n)) return C_KZG_BADARGS;
    G1Xyzz acc = G1Xyzz::infinity();
    for (size_t i = 0; i < n; i++) {
        G1Affine p;
        p.x = Fp::zero();
        p.y = Fp::zero();
        int rc = g1_decompress_nocheck(p, points48 + 48 * i);
        if (rc == 2) {
            set_error("lwkzg_g1_sum_compressed: point %zu is not a valid compressed G1 point", i);
            return C_KZG_BADARGS;
        }
        if (rc == 0) acc = xyzz_madd(acc, p);
    }
    g1_compress(out48, acc);
    return C_KZG_OK;
}

// test hook: the host-side Fiat-Shamir digests (SHA extensions when the CPU has them), no GPU
C_KZG_RET lwkzg_challenge_digests_host(uint8_t *digests32, const uint8_t *blobs, const uint8_t *commitments48, size_t n) {
    if (!digests32 || !blobs || !commitments48) return C_KZG_BADARGS;
    challenge_digests_host(digests32, blobs, commitments48, n);
    return C_KZG_OK;
}

// the batch challenge r over a transcript of records, canonical big-endian (host only)
C_KZG_RET lwkzg_batch_challenge_host(uint8_t r_out[32], const uint8_t *records_all, size_t n_total, int mode) {
    if (!r_out || (!records_all && n_total) || (mode != LWKZG_MODE_REFERENCE && mode != LWKZG_MODE_CKZG)) return C_KZG_BADARGS;
    const uint32_t one_limbs[8] = {1, 0, 0, 0, 0, 0, 0, 0};
    hfr_to_be(r_out, batch_challenge_mont(records_all, n_total, mode == LWKZG_MODE_CKZG) * hfr_raw(one_limbs));
    return C_KZG_OK;
}

// test hook: prod e(P_i, Q_i) == 1 on compressed inputs, host only (no GPU, no settings)
C_KZG_RET lwkzg_pairing_product_is_one(bool *ok, const uint8_t *g1_compressed, const uint8_t *g2_compressed, size_t n) {
    if (!ok || n > 4) return C_KZG_BADARGS;
    *ok = false;
    return pairing_check_compressed(g1_compressed, g2_compressed, (int)n, ok) ? C_KZG_OK : C_KZG_BADARGS;
}
}
