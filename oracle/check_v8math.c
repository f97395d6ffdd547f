/*
 * TEST INFRASTRUCTURE: pins oracle/v8math.h against the routines V8 itself uses.
 * dlopen()s libnode.so.72 (present in this image) and compares v8_log/v8_log10/v8_pow
 * bit-for-bit with v8::base::ieee754::{log,log10,pow} on N random samples per domain.
 * build+run: make -C oracle check_v8math
 */
#include <stdio.h>
#include <stdlib.h>
#include <dlfcn.h>
#include "v8math.h"

static uint64_t rs = 88172645463325252ULL;
static uint64_t xr(void) { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return rs; }
static double u01(void) { return (double)(xr() >> 11) / 9007199254740992.0; }

int main(int argc, char** argv) {
    long N = argc > 1 ? atol(argv[1]) : 2000000;
    void* h = dlopen("libnode.so.72", RTLD_NOW | RTLD_GLOBAL);
    if (!h) { printf("SKIP: libnode.so.72 not loadable: %s\n", dlerror()); return 77; }
    double (*rlog)(double) = (double (*)(double))dlsym(h, "_ZN2v84base7ieee7543logEd");
    double (*rlog10)(double) = (double (*)(double))dlsym(h, "_ZN2v84base7ieee7545log10Ed");
    double (*rpow)(double, double) = (double (*)(double, double))dlsym(h, "_ZN2v84base7ieee7543powEdd");
    if (!rlog || !rlog10 || !rpow) { printf("SKIP: symbols missing\n"); return 77; }
    long bad = 0;
    for (long i = 0; i < N; i++) {
        /* log/log10 over ~[1e-30, 1e30] log-uniform plus raw bit patterns */
        double x = exp((u01() - 0.5) * 140.0);
        if (i % 7 == 0) { uint64_t b = xr() & 0x7fefffffffffffffULL; memcpy(&x, &b, 8); }
        double a = v8_log(x), b = rlog(x);
        if (memcmp(&a, &b, 8)) { if (bad < 10) printf("log(%a): %a vs %a\n", x, a, b); bad++; }
        a = v8_log10(x); b = rlog10(x);
        if (memcmp(&a, &b, 8)) { if (bad < 10) printf("log10(%a): %a vs %a\n", x, a, b); bad++; }
        /* pow(10, y), y in [-40, 40] (athAdjust range and beyond) */
        double y = (u01() - 0.5) * 80.0;
        a = v8_pow(10.0, y); b = rpow(10.0, y);
        if (memcmp(&a, &b, 8)) { if (bad < 10) printf("pow(10,%a): %a vs %a\n", y, a, b); bad++; }
        /* pow(x, .5) == sqrt; pow(2, k/16); general */
        a = v8_pow(x, 0.5); b = rpow(x, 0.5);
        if (memcmp(&a, &b, 8)) { if (bad < 10) printf("pow(%a,.5): %a vs %a\n", x, a, b); bad++; }
        double xx = u01() * 100.0, yy = (u01() - 0.5) * 20.0;
        a = v8_pow(xx, yy); b = rpow(xx, yy);
        if (memcmp(&a, &b, 8)) { if (bad < 10) printf("pow(%a,%a): %a vs %a\n", xx, yy, a, b); bad++; }
    }
    printf("%s: %ld samples x5 functions, %ld mismatches\n", bad ? "FAIL" : "OK", N, bad);
    return bad ? 1 : 0;
}
