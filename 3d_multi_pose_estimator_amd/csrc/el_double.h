// Decimal -> binary64, correctly rounded, for the number shapes of the frame JSON (host and device).
//
// The device-side parser (jsonparse.hip) converts the doubles of the wire format itself.  Python's
// float() -- what the reference's json.loads applies to every number (metrics_from_model.py:182-191) --
// is correctly rounded, so the conversion has to be exact, not "close".  Algorithm: Eisel-Lemire
// (D. Lemire, "Number parsing at a gigabyte per second", Software: Practice and Experience 51(8),
// 2021): the decimal significand w (<= 19 digits, exact in 64 bits) times a 128-bit truncated
// significand of 5^q decides the rounding of w * 10^q whenever the product is not within one unit of a
// rounding boundary; for q in [-27, 55] the second 64 x 64 product settles every remaining case
// (section 8 of the paper), so inside that range the result is ALWAYS the correctly rounded one and
// outside it -- or with more than 19 significant digits, or results outside the normal range -- the
// function declines (returns false) and the caller hands the token to the host (glibc strtod).
// tests/test_host_logic.py::test_eisel_lemire_against_strtod checks it against strtod on random and
// adversarial tokens through the host build of this same header.
#pragma once
#include <stdint.h>

#include "pow5_table.h"

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define MPE_HD __host__ __device__
#else
#define MPE_HD
#endif

namespace mpe {

#if defined(__HIP_DEVICE_COMPILE__)
__device__ static const uint64_t el_pow5[] = MPE_POW5_TABLE;
#else
static const uint64_t el_pow5[] = MPE_POW5_TABLE;
#endif

MPE_HD inline void el_mul64(uint64_t a, uint64_t b, uint64_t *hi, uint64_t *lo) {
#if defined(__HIP_DEVICE_COMPILE__)
    *lo = a * b;
    *hi = __umul64hi(a, b);
#else
    const unsigned __int128 p = (unsigned __int128)a * b;
    *lo = (uint64_t)p;
    *hi = (uint64_t)(p >> 64);
#endif
}

MPE_HD inline int el_clz64(uint64_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __clzll((long long)x);
#else
    return __builtin_clzll(x);
#endif
}

// w * 10^q -> nearest binary64 (ties to even).  w != 0, at most 19 decimal digits (exact).  false =
// not decided here (q outside [MPE_EL_QMIN, MPE_EL_QMAX], or a subnormal / overflowing result).
MPE_HD inline bool el_to_double(uint64_t w, int q, bool negative, double *out) {
    if (w == 0 || q < MPE_EL_QMIN || q > MPE_EL_QMAX) return false;
    int lz = el_clz64(w);
    w <<= lz;
    const int idx = 2 * (q - MPE_EL_QMIN);
    uint64_t hi, lo;
    el_mul64(w, el_pow5[idx], &hi, &lo);
    // precision needed: 52 explicit significand bits + 3; if the low 9 bits of `hi` are all ones the
    // truncated table entry may have cut a carry: refine with the second word of the table
    if ((hi & 0x1FFu) == 0x1FFu) {
        uint64_t h2, l2;
        el_mul64(w, el_pow5[idx + 1], &h2, &l2);
        lo += h2;
        if (h2 > lo) ++hi;
    }
    const int upperbit = (int)(hi >> 63);
    uint64_t mant = hi >> (upperbit + 64 - 52 - 3);
    // biased binary exponent: floor(q * log2(10)) = (217706 * q) >> 16 for |q| <= 350; + 63 for the normalised w,
    // + 1023 bias (the paper's power(q) + upperbit - lz - minimum_exponent)
    int64_t power2 = (((int64_t)(152170 + 65536) * q) >> 16) + 63 + upperbit - lz + 1023;
    if (power2 <= 0 || power2 >= 0x7FF) return false;           // subnormal or overflow: not handled here
    // exactly halfway between two doubles can only happen for small |q| (5^q fits 64 bits): round to even
    if (lo <= 1 && q >= -4 && q <= 23 && (mant & 3) == 1) {
        if ((mant << (upperbit + 64 - 52 - 3)) == hi) mant &= ~(uint64_t)1;
    }
    mant += mant & 1;
    mant >>= 1;
    if (mant >= ((uint64_t)2 << 52)) {
        mant = (uint64_t)1 << 52;
        ++power2;
        if (power2 >= 0x7FF) return false;
    }
    mant &= ~((uint64_t)1 << 52);
    uint64_t bits = mant | ((uint64_t)power2 << 52) | (negative ? (uint64_t)1 << 63 : 0);
    union {
        uint64_t u;
        double d;
    } cv;
    cv.u = bits;
    *out = cv.d;
    return true;
}

// One JSON number token: [-+]digits[.digits][(e|E)[-+]digits], read through a byte source `Src` with
//   int peek(int k)   byte at the cursor + k (0 beyond the end)      void skip(int k)   advance the cursor
// Returns the number of bytes consumed (0 = no number here; the cursor then has not moved) and sets *ok = false
// when the conversion is left to the host (more than 19 significant digits, exponent out of range, ...); *v is
// exact when *ok.
template <class Src>
MPE_HD inline int el_parse_number_src(Src &src, double *v, bool *ok) {
    int q = 0;
    bool neg = false;
    int c = src.peek(0);
    if (c == '-' || c == '+') {
        neg = c == '-';
        ++q;
    }
    uint64_t m = 0;
    int digits = 0, exp10 = 0;
    bool dropped = false, any = false;
    for (c = src.peek(q); c >= '0' && c <= '9'; c = src.peek(++q)) {
        if (digits < 19) {
            m = m * 10 + (uint64_t)(c - '0');
            if (m) ++digits;
        } else {
            ++exp10;
            dropped = dropped || c != '0';
        }
        any = true;
    }
    if (c == '.') {
        for (c = src.peek(++q); c >= '0' && c <= '9'; c = src.peek(++q)) {
            if (digits < 19) {
                m = m * 10 + (uint64_t)(c - '0');
                if (m) ++digits;
                --exp10;
            } else {
                dropped = dropped || c != '0';
            }
            any = true;
        }
    }
    if (!any) return 0;
    if (c == 'e' || c == 'E') {
        int es = q + 1;
        bool eneg = false;
        int d = src.peek(es);
        if (d == '-' || d == '+') {
            eneg = d == '-';
            ++es;
        }
        int ev = 0;
        const int e0 = es;
        for (d = src.peek(es); d >= '0' && d <= '9' && ev < 10000; d = src.peek(++es)) ev = ev * 10 + (d - '0');
        if (es > e0) {
            exp10 += eneg ? -ev : ev;
            q = es;
        }
    }
    *ok = true;
    if (m == 0) {
        *v = neg ? -0.0 : 0.0;
    } else if (dropped) {
        *ok = false;
    } else if (exp10 == 0 && m <= ((uint64_t)1 << 53)) {
        *v = neg ? -(double)m : (double)m;                       // small integers: exact
    } else if (!el_to_double(m, exp10, neg, v)) {
        *ok = false;
    }
    src.skip(q);
    return q;
}

struct ElPtrSrc {
    const char *p, *end;
    MPE_HD int peek(int k) const { return p + k < end ? (unsigned char)p[k] : 0; }
    MPE_HD void skip(int k) { p += k; }
};

MPE_HD inline int el_parse_number(const char *p, const char *end, double *v, bool *ok) {
    ElPtrSrc s{p, end};
    return el_parse_number_src(s, v, ok);
}

}  // namespace mpe
