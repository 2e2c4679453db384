/* Plain-C consumer of include/pagk.h: proves the boundary is a C ABI (no C++ / torch types).
 * Built and run by tests/test_capi_cpu.py.  Exit code 0 = every check passed. */
#include <stdio.h>
#include <string.h>

#include "pagk.h"

int main(void)
{
    pagk_params p;
    pagk_ctx *ctx = NULL;
    int rc;
    if (pagk_version() != PAGK_VERSION) return 1;
    pagk_params_default(&p);
    if (p.half_patch != 5 || p.iterations != 10 || p.pyramids != 3 || !p.consider_affine) return 2;
    if (strcmp(pagk_strerror(PAGK_E_UNSUPPORTED), "unsupported mode") != 0) return 3;
    rc = pagk_create(&ctx, 0);
    if (rc == PAGK_OK) {
        /* a device is present: exercise argument checking, then clean up */
        if (pagk_track(ctx, &p, NULL, NULL, 0, NULL, NULL, NULL, NULL, NULL) != PAGK_E_ARG) return 4;
        p.inverse = 1;
        {
            unsigned char px[4 * 4] = {0};
            pagk_image im = {px, 4, 4, 4};
            float pt[2] = {1, 1}, A[4] = {1, 0, 0, 1};
            unsigned char st = 1, so = 0;
            float out_pt[2];
            pagk_outputs o;
            memset(&o, 0, sizeof o);
            o.pt_un = out_pt;
            o.status = &so;
            if (pagk_track(ctx, &p, &im, &im, 1, pt, pt, A, &st, &o) != PAGK_E_UNSUPPORTED) return 5;
        }
        pagk_destroy(ctx);
        printf("device present\n");
    } else if (rc == PAGK_E_NODEVICE) {
        if (ctx != NULL) return 6;
        printf("no device: PAGK_E_NODEVICE, no fallback\n");
    } else {
        return 7;
    }
    {
        /* the host-side consumer row works without a device */
        unsigned char st[2] = {1, 0}, out[2];
        double err[2] = {1.0, 1.0}, dist[2] = {1.0, 1.0};
        if (pagk_post_filter(2, 5, st, err, dist, NULL, NULL, out, NULL, NULL) != 1 || out[0] != 1 || out[1] != 0) return 8;
    }
    return 0;
}
