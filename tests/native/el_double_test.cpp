// Host build of csrc/el_double.h (the decimal -> binary64 conversion of the device-side JSON parser) against glibc
// strtod: every token the fast path accepts must give strtod's bits; tokens it declines are only counted.
//   g++ -O2 -I 3d_multi_pose_estimator_amd/csrc tests/native/el_double_test.cpp -o el_double_test && ./el_double_test [iterations]
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include "el_double.h"      // -I <repo>/3d_multi_pose_estimator_amd/csrc
int main(int argc, char **argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 3000000;
    std::mt19937_64 rng(12345);
    long tested = 0, declined = 0, bad = 0;
    char buf[128];
    auto check = [&](const char *s) {
        double v = 0; bool ok = false;
        int n = mpe::el_parse_number(s, s + strlen(s), &v, &ok);
        char *e; double r = strtod(s, &e);
        ++tested;
        if (n != (int)(e - s)) { if (bad < 10) printf("LEN %s: %d vs %d\n", s, n, (int)(e - s)); ++bad; return; }
        if (!ok) { ++declined; return; }
        if (memcmp(&v, &r, 8) != 0) { if (bad < 10) printf("BAD %s: %.17g vs %.17g\n", s, v, r); ++bad; }
    };
    // python-repr-like doubles (pixel coordinates), random full-precision doubles, 19-digit mantissas, halfway cases
    for (int i = 0; i < iters; ++i) {
        double x = std::ldexp((double)(rng() >> 11), -53) * 1920.0;
        snprintf(buf, sizeof buf, "%.17g", x); check(buf);
        snprintf(buf, sizeof buf, "%.15g", x); check(buf);
        uint64_t u = rng(); double y; memcpy(&y, &u, 8);
        if (y == y && std::abs(y) < 1e300 && std::abs(y) > 1e-300) { snprintf(buf, sizeof buf, "%.17g", y); check(buf); snprintf(buf, sizeof buf, "%.17e", y); check(buf); }
        uint64_t m = rng() % 10000000000000000000ULL; int q = (int)(rng() % 90) - 35;
        snprintf(buf, sizeof buf, "%llue%d", (unsigned long long)m, q); check(buf);
        // near-halfway: a double plus/minus half an ulp printed with 19 digits
        double z = std::ldexp((double)(rng() >> 11) + 0.0, -20 - (int)(rng() % 30));
        snprintf(buf, sizeof buf, "%.19g", z); check(buf);
    }
    const char *fixed[] = {"0", "-0", "0.0", "1", "-1", "17", "0.5", "960.0", "1e22", "1e23", "8.5e-5", "9007199254740993", "9007199254740992.5", "0.1", "0.30000000000000004",
                           "123456789012345678901234567890", "1.7976931348623157e308", "5e-324", "2.2250738585072014e-308", "1e-27", "1e55", "1e56", "4.35", "1e-28"};
    for (const char *f : fixed) check(f);
    printf("tested %ld declined %ld bad %ld\n", tested, declined, bad);
    return bad != 0;
}
