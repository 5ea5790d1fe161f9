// Timings of the HOST field tower under the pairing (hostfp.h, fp_x86.S, pairing.hip), on whatever core runs it; no GPU involved.
//   tools/host_field_bench.sh            builds it against the library's objects and runs it
//   LWKZG_EXPERIMENTAL=1 LWKZG_HOST_FP_PORTABLE=1|2 ...   the C products (1), or only the Fp2 product in C (2)
// pairing.hip is included as a whole: its tower lives in an anonymous namespace.
#include "../lambdaworks_kzg_amd/csrc/pairing.hip"
using namespace lwk;
template <class F> double timeit(F f, int n) {
    double best = 1e30;
    for (int rep = 0; rep < 7; rep++) {
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < n; i++) f();
        auto t1 = std::chrono::steady_clock::now();
        double v = std::chrono::duration<double, std::nano>(t1 - t0).count() / n;
        if (v < best) best = v;
    }
    return best;
}
int main() {
    init_consts();
    HFp s = HFp::one(); s.l[0] ^= 0x123456789; s = s * s;
    auto nx = [&]() { s = s * s + HFp::one(); return s; };
    H12 a, b;
    H2 *pa = &a.c0.c0, *pb = &b.c0.c0;
    for (int i = 0; i < 6; i++) { pa[i] = {nx(), nx()}; pb[i] = {nx(), nx()}; }
    HFp x = nx(), y = nx();
    H2 u = {nx(), nx()}, v = {nx(), nx()};
    printf("Fp mul      %8.1f ns\n", timeit([&]() { x = x * y; }, 200000));
    printf("Fp add      %8.1f ns\n", timeit([&]() { x = x + y; }, 200000));
    printf("Fp sub      %8.1f ns\n", timeit([&]() { x = x - y; }, 200000));
    printf("Fp2 mul     %8.1f ns\n", timeit([&]() { u = u * v; }, 100000));
    printf("Fp2 mul_xi  %8.1f ns\n", timeit([&]() { u = mul_xi(u); }, 100000));
    printf("Fp6 mul     %8.1f ns\n", timeit([&]() { a.c0 = a.c0 * b.c0; }, 20000));
    printf("Fp12 mul    %8.1f ns\n", timeit([&]() { a = a * b; }, 10000));
    printf("Fp12 sqr    %8.1f ns\n", timeit([&]() { a = f12sqr(a); }, 10000));
    printf("cyc sqr     %8.1f ns\n", timeit([&]() { a = cyclotomic_sqr(a); }, 10000));
    printf("mul_by_line %8.1f ns\n", timeit([&]() { a = f12mul_by_line(a, u, v, x); }, 10000));
    printf("f12inv      %8.1f ns\n", timeit([&]() { a = f12inv(a); }, 1000));
    printf("frob_p      %8.1f ns\n", timeit([&]() { a = frob_p(a); }, 10000));
    H12 t = f12conj(a) * f12inv(a); t = frob_p2(t) * t;
    printf("exp_by_x    %8.1f ns\n", timeit([&]() { t = exp_by_x(t); }, 200));
    bool r = false;
    printf("final exp   %8.1f ns\n", timeit([&]() { r ^= final_exponentiation_is_one(a); }, 100));
    printf("%d %llx\n", (int)r, (unsigned long long)(x.l[0] ^ u.c0.l[0] ^ a.c0.c0.c0.l[0] ^ t.c0.c0.c0.l[0]));
}
