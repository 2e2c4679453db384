/* tests/repeat_sum_check.c -- host restatement of repeat_sum_f64 (csrc/pagk_device.h), operation for operation, checked
 * against the loop it replaces: s_0 = 0, s_k = RN(s_{k-1} + c*c), k = 1..P  (H(2,2) of src/patch_match.cpp:296 with
 * J[2] = c constant over the patch, :263).  Test infrastructure (tests/test_repeat_sum.py builds and runs it).
 *   repeat_sum_check <random cases> <exhaustive: 0|1> <scale of the quotient estimate, e.g. 1.0 / 1.00003 / 0.99997>
 * exit code 0 = every case bit-identical. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static inline uint64_t bits(double x) { uint64_t u; memcpy(&u, &x, 8); return u; }
static inline double frombits(uint64_t u) { double x; memcpy(&x, &u, 8); return x; }

static double loop_sum(double q, int P)
{
    double s = 0;
    for (int k = 0; k < P; k++) s = s + q; /* q = c*c is exact in double: fma(c, c, s) == RN(s + q) */
    return s;
}

static float g_scale = 1.0f;
static double closed_sum(double q, int P)
{
    if (!(q > 0.0)) return (double)P * q;
    double x = 32.0 * q;
    int k = 32;
    double top = frombits((((bits(q) >> 52) & 0x7ff) + 6) << 52); /* 2^(E+6) */
    double M = 0.75 * top;                                          /* 1.5 * 2^(E+5) */
    for (int b = 0; b < 5; b++) {
        const double I = (q + M) - M;
        const float est = (float)(top - x) * (1.0f / (float)I) * g_scale; /* the device uses v_rcp_f32 (1 ulp) */
        int n = (int)est - 1;
        const int lim = (P - k) - 4 * (5 - b);
        n = n > lim ? lim : n;
        n = n < 0 ? 0 : n;
        x = fma((double)n, I, x);
        k += n + 4;
        x = x + q; x = x + q; x = x + q; x = x + q;
        M *= 2.0;
        top *= 2.0;
    }
    if (k != P) { fprintf(stderr, "step count %d != %d\n", k, P); exit(2); }
    return x;
}

static uint64_t splitmix(uint64_t *s)
{
    uint64_t z = (*s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

int main(int argc, char **argv)
{
    const long N = argc > 1 ? atol(argv[1]) : 1000000;
    const int exhaustive = argc > 2 ? atoi(argv[2]) : 0;
    if (argc > 3) g_scale = (float)atof(argv[3]);
    const int Ps[3] = {289, 361, 441}; /* h = 8, 9, 10 */
    uint64_t s = 12345;
    long bad = 0;
    for (long t = 0; t < N; t++) {
        const uint64_t r = splitmix(&s);
        uint32_t m = (uint32_t)(r & 0x7fffff), ex;
        const int mode = (int)(t % 4);
        if (mode == 0) ex = 127 + (r >> 40) % 9;                 /* [1, 256): a pixel value */
        else if (mode == 1) ex = 127 - (r >> 40) % 60;           /* tiny interpolation products */
        else if (mode == 2) { ex = 127 + (r >> 40) % 9; m &= ~((1u << ((r >> 50) % 23)) - 1); } /* trailing zeros */
        else { ex = 127 + (r >> 40) % 9; m |= (r >> 33) & 1 ? 0x7ffff0 : 0; }                   /* runs of ones */
        const uint32_t u = (ex << 23) | m;
        float c;
        memcpy(&c, &u, 4);
        const double q = (double)c * (double)c;
        for (int pi = 0; pi < 3; pi++)
            if (bits(loop_sum(q, Ps[pi])) != bits(closed_sum(q, Ps[pi]))) {
                if (bad < 10) printf("MISMATCH c=%a P=%d\n", c, Ps[pi]);
                bad++;
            }
    }
    if (exhaustive)
        for (uint32_t m = 0; m < (1u << 23); m++) { /* every mantissa at one exponent: the sums are scale-invariant */
            const uint32_t u = (130u << 23) | m;
            float c;
            memcpy(&c, &u, 4);
            const double q = (double)c * c;
            for (int pi = 0; pi < 3; pi++)
                if (bits(loop_sum(q, Ps[pi])) != bits(closed_sum(q, Ps[pi]))) {
                    if (bad < 10) printf("MISMATCH c=%a P=%d (exhaustive)\n", c, Ps[pi]);
                    bad++;
                }
        }
    if (bits(closed_sum(0.0, 441)) != bits(0.0)) bad++;
    printf("%ld random cases x 3 patch sizes%s, estimate scale %g: %ld mismatches\n", N, exhaustive ? " + 2^23 mantissas" : "", (double)g_scale, bad);
    return bad != 0;
}
