/* The host twin of beam_expf_neg / beam_log1pf_unit (csrc/beam_kernels.hip): every third float in [0, 17)
 * against glibc (float)exp((double)) / (float)log1p((double)).  Expected: 0 mismatches of 366 M. */
#include <math.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include "beam_lse_tables.h"   /* python tools/beam_lse_tables.py > /tmp/beam_lse_tables.h; gcc -I/tmp ... */
static float expf_neg(float a) {
    const double x = -(double)a;
    const double nf = rint(x * INV_LN2_32);
    double r = fma(nf, -LN2_32_HI, x);
    r = fma(nf, -LN2_32_LO, r);
    const int n = (int)nf;
    const int j = n & 31, k = n >> 5;
    double p = 1.0 / 720.0;
    p = fma(p, r, 1.0 / 120.0);
    p = fma(p, r, 1.0 / 24.0);
    p = fma(p, r, 1.0 / 6.0);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return (float)ldexp(EXP2_32[j] * p, k);
}
static float log1pf_unit(float e) {
    const double t = 1.0 + (double)e;
    const int i = (int)rint((t - 1.0) * 64.0);
    const double r = fma(t, INVC[i], -1.0);
    double q = 1.0 / 7.0;          /* r - r^2/2 + ... + r^7/7 */
    q = fma(q, -r, 1.0 / 6.0);
    q = fma(q, -r, 1.0 / 5.0);
    q = fma(q, -r, 1.0 / 4.0);
    q = fma(q, -r, 1.0 / 3.0);
    q = fma(q, -r, 0.5);
    q = fma(q, -r, 1.0);
    return (float)fma(q, r, LOGC[i]);
}
int main() {
    long bad_e = 0, bad_l = 0, n = 0;
#ifndef STRIDE
#define STRIDE 3      /* every third float in [0, 17): 366 M arguments, ~16 s */
#endif
    for (uint32_t u = 0; u < 0x41880000u; u += STRIDE) {
        float a; memcpy(&a, &u, 4);
        const float e_ref = (float)exp(-(double)a), e = expf_neg(a);
        if (e != e_ref) { if (bad_e < 5) printf("exp a=%a got %a want %a\n", a, e, e_ref); ++bad_e; }
        const float l_ref = (float)log1p((double)e_ref), l = log1pf_unit(e_ref);
        if (l != l_ref) { if (bad_l < 5) printf("log1p e=%a got %a want %a\n", e_ref, l, l_ref); ++bad_l; }
        ++n;
    }
    printf("checked %ld arguments: expf mismatches %ld, log1pf mismatches %ld\n", n, bad_e, bad_l);
    return 0;
}
