// Test harness (CPU): fray_amd/csrc/dev_trig.hpp against 113-bit arithmetic (libquadmath) and against this host's glibc, over the
// arguments the sampler produces: theta = 2 pi u, phi = acos(2 v - 1), acos(2 v - 1).  Prints the mismatch counts; exits 1 when
// the functions are not correctly rounded in all but 2 per 10^4 calls.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <quadmath.h>
#define FRAY_TRIG_FN static inline
#define FRAY_TRIG_TABLE static const
#define FRAY_TRIG_LIBM_SINCOS(x, s, c) ::sincos(x, s, c)
#define FRAY_TRIG_LIBM_ACOS(x) ::acos(x)
#include "dev_trig.hpp"

static uint64_t st = 88172645463325252ULL;
static double u01() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (st >> 11) * (1.0 / 9007199254740992.0); }

int main(int argc, char** argv)
{
    const long n = argc > 1 ? atol(argv[1]) : 3000000;
    long crS = 0, crC = 0, crA = 0, glS = 0, glC = 0, glA = 0, g_crS = 0, g_crC = 0, g_crA = 0, fused = 0;
    for (long i = 0; i < n; i++) {
        const double v = 2 * u01() - 1;
        const double x = (i & 1) ? 2 * 3.141592653589793238 * u01() : ::acos(v);      // theta, or phi
        double s, c;
        fray_sincos(x, &s, &c);
        const double a = fray_acos(v);
        const double qs = (double)sinq((__float128)x), qc = (double)cosq((__float128)x), qa = (double)acosq((__float128)v);
        crS += s != qs; crC += c != qc; crA += a != qa;
        {   // the fused form the hemisphere sampler uses must give what the two calls give
            double fs, fc, ts, tc;
            fray_acos_sincos(v, &fs, &fc);
            fray_sincos(a, &ts, &tc);
            fused += fs != ts || fc != tc;
        }
        glS += s != ::sin(x); glC += c != ::cos(x); glA += a != ::acos(v);
        g_crS += ::sin(x) != qs; g_crC += ::cos(x) != qc; g_crA += ::acos(v) != qa;
    }
    printf("samples %ld\n", n);
    printf("dev_trig vs correctly rounded : sin %ld cos %ld acos %ld\n", crS, crC, crA);
    printf("dev_trig vs this host's glibc : sin %ld cos %ld acos %ld\n", glS, glC, glA);
    printf("glibc    vs correctly rounded : sin %ld cos %ld acos %ld\n", g_crS, g_crC, g_crA);
    printf("fray_acos_sincos vs fray_sincos(fray_acos) : %ld differ\n", fused);
    const long lim = n / 5000 + 1;
    return (crS <= lim && crC <= lim && crA <= lim && fused <= lim) ? 0 : 1;
}
