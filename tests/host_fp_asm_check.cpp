// Harness of tests/test_host_fp_asm_cpu.py: reads pairs of six-limb operands (hex, least significant limb first), prints the
// product fp_x86.S computes for each. First line of output: whether this core has BMI2 + ADX (0: nothing else is printed).
#include <cstdint>
#include <cstdio>
extern "C" void lwk_fp_mul_adx(uint64_t *r, const uint64_t *a, const uint64_t *b);
extern "C" int lwk_cpu_has_bmi2_adx(void);
int main() {
    const int have = lwk_cpu_has_bmi2_adx();
    printf("%d\n", have);
    if (!have) return 0;
    unsigned long a[6], b[6];
    while (scanf("%lx %lx %lx %lx %lx %lx %lx %lx %lx %lx %lx %lx", a, a + 1, a + 2, a + 3, a + 4, a + 5, b, b + 1, b + 2, b + 3, b + 4, b + 5) == 12) {
        uint64_t x[6], y[6], r[6];
        for (int k = 0; k < 6; k++) x[k] = a[k], y[k] = b[k];
        lwk_fp_mul_adx(r, x, y);
        lwk_fp_mul_adx(x, x, y);   // the result over its first operand, as `a = a * b` does
        for (int k = 0; k < 6; k++)
            if (x[k] != r[k]) return 3;
        printf("%lx %lx %lx %lx %lx %lx\n", (unsigned long)r[0], (unsigned long)r[1], (unsigned long)r[2], (unsigned long)r[3], (unsigned long)r[4], (unsigned long)r[5]);
    }
    return 0;
}
