#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include "pagk.h"
#include "pagk_oracle.h"
/* Sanitizer driver (built with -fsanitize=address,undefined by tests/test_oracle.py): the oracle's entry
 * points on small random inputs -- odd sizes, padded rows, points on the border, far outside and NaN. */
static unsigned long long s = 88172645463325252ull;
static double rnd(void) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (s >> 11) * (1.0 / 9007199254740992.0); }
int main(void)
{
    for (int trial = 0; trial < 40; trial++) {
        int w = 20 + (int)(rnd() * 150), h = 16 + (int)(rnd() * 120), L = 1 + (int)(rnd() * 4), n = (int)(rnd() * 60);
        int hp = 1 + (int)(rnd() * 12);
        int step = w + (int)(rnd() * 5);
        unsigned char *a = malloc((size_t)step * h), *b = malloc((size_t)step * h);
        for (int i = 0; i < step * h; i++) a[i] = (unsigned char)(rnd() * 256), b[i] = (unsigned char)(rnd() * 256);
        pagk_image ia = {a, w, h, step}, ib = {b, w, h, step};
        pagk_params p;
        memset(&p, 0, sizeof p); /* the call site's constants, src/gyro_aided_tracker.cpp:276-282, src/patch_match.cpp:48-50 */
        p.lambda = 1.0f, p.alpha = 0.5f, p.max_distance = 25, p.n_dist_coef = 4;
        p.half_patch = hp; p.iterations = 1 + (int)(rnd() * 12); p.pyramids = L;
        p.consider_affine = rnd() < 0.5; p.consider_illumination = rnd() < 0.5; p.regularization_penalty = rnd() < 0.5;
        p.calculate_ncc = rnd() < 0.3; p.has_gyro_predict_initial = rnd() < 0.7;
        p.fx = 300; p.fy = 310; p.cx = w / 2.0f; p.cy = h / 2.0f; p.dist_coef[0] = rnd() < 0.5 ? 0.1f : 0.0f;
        p.inv_log_max_dist = pagk_oracle_inv_log_max_dist(p.alpha, p.max_distance);
        int nn = n > 0 ? n : 1;
        float *pr = malloc(8 * nn), *pi = malloc(8 * nn), *af = malloc(16 * nn), *pu = malloc(8 * nn), *pd = malloc(8 * nn), *ncc = malloc(4 * nn);
        unsigned char *st = malloc(nn), *so = malloc(nn);
        double *pe = malloc(8 * nn), *dp = malloc(8 * nn); int *it = malloc(4 * nn);
        for (int i = 0; i < n; i++) {
            /* points inside, on the border, far outside, NaN */
            double r = rnd();
            pr[2*i] = (float)(r < 0.8 ? rnd() * w : (r < 0.9 ? -50 + rnd() * (w + 100) : (r < 0.95 ? 1e9 : NAN)));
            pr[2*i+1] = (float)(rnd() * h);
            pi[2*i] = pr[2*i] + (float)(rnd() * 4 - 2); pi[2*i+1] = pr[2*i+1] + (float)(rnd() * 4 - 2);
            af[4*i] = 1 + (float)(rnd() * 0.1); af[4*i+1] = (float)(rnd() * 0.1); af[4*i+2] = (float)(rnd() * 0.1); af[4*i+3] = 1;
            st[i] = rnd() < 0.9;
        }
        pagk_outputs o = {pu, pd, so, pe, dp, ncc, it};
        int rc = pagk_oracle_track(&p, &ia, &ib, n, pr, pi, af, st, &o, 1 + (int)(rnd() * 3));
        if (rc != PAGK_OK && rc != PAGK_E_ARG) { printf("trial %d rc %d\n", trial, rc); return 1; }
        /* post filter, predict, geometry */
        float *pp = malloc(8 * nn), *ppu = malloc(8 * nn);
        if (rc == PAGK_OK) pagk_oracle_post_filter(n, hp, so, pe, dp, pd, pu, st, pp, ppu);
        float KRK[9] = {1, 0.001f, 2, -0.001f, 1, -1, 1e-6f, 0, 1}, r3[3] = {0.001f, -0.002f, 1};
        pagk_oracle_gyro_predict(&p, w, h, hp, KRK, r3, n, pr, pu, pd, so, af);
        double H[9] = {1, 0, 1.5, 0, 1, -0.5, 0, 0, 1}, Hi[9] = {1, 0, -1.5, 0, 1, 0.5, 0, 0, 1}, F[9] = {0, 0, 0, 0, 0, -1, 0, 1, 0};
        float sc;
        pagk_oracle_geometry_validation(H, Hi, F, n, pr, pi, st, 1.0f, &sc);
        free(a); free(b); free(pr); free(pi); free(af); free(pu); free(pd); free(ncc); free(st); free(so); free(pe); free(dp); free(it); free(pp); free(ppu);
    }
    printf("asan driver ok\n");
    return 0;
}
