// Host-side formatter of the dataset CSV rows (src/s01-dataset-generation.py:218-241): one row per particle per
// state, the nine fp32 state columns printed exactly as Python's csv module prints the numpy.float32 scalars the
// reference hands it, i.e. str(np.float32(x)): the shortest decimal that reads back as the same fp32 (rounding
// interval bounds count when the mantissa is even), nearest to the value among those, positional for
// 1e-4 <= |x| < 1e16 and d.ddde-XX otherwise. No device code in this file: the rows are formatted on the host from
// the state snapshots run() already brought back; it lives in the C-ABI library because the per-value work (about
// 1.5 us in numpy's generic formatter, nine values per row) is what the dataset CLI spends its time on once the
// integrator runs on the GPU.
//
// The digit generation is exact integer arithmetic, not a table-approximated scheme: an fp32 and its two interval
// bounds are (4m, 4m+2, 4m-1 or 4m-2) * 2^e; scaled by a power of ten they are either 128-bit integers divided by
// 10^q (e >= 0) or a 26-bit x 107-bit product shifted right (e < 0), so floor, remainder, "is exactly an integer"
// and the first dropped digit are all exact. The digit-removal loop on the three scaled integers follows the
// structure of Adams, "Ryu: fast float-to-string conversion" (PLDI 2018), section 3.
// tests/test_csv_format.py compares against numpy on every exponent's boundary values and on random bit patterns;
// tools/check_f32_format_exhaustive.py runs all 2^32 patterns.
#include <stdint.h>
#include <string.h>

#include "../../include/nbd.h"

namespace {

typedef unsigned __int128 u128;

struct Tables {
    u128 pow5[48];
    u128 pow10[32];
    Tables() {
        pow5[0] = 1;
        for (int i = 1; i < 48; ++i) pow5[i] = pow5[i - 1] * 5;      // 5^47 < 2^110
        pow10[0] = 1;
        for (int i = 1; i < 32; ++i) pow10[i] = pow10[i - 1] * 10;    // 10^31 < 2^103
    }
};
const Tables kT;

struct Scaled { uint32_t q; uint32_t first; bool exact; bool below_first_zero; };
// q = floor(value), exact = value is an integer, first = first digit after the point,
// below_first_zero = nothing but zeros after that digit

// floor(m * 2^e2 / 10^k) for e2 >= 0 (k = floor(e2 log10 2), so the quotient is below 10 * 2^26)
inline Scaled scale_up(uint32_t m, int e2, int k) {
    const u128 num = (u128)m << e2, den = kT.pow10[k];
    const u128 q = num / den, r = num - q * den;
    const u128 t = r * 10, d = t / den;
    return {(uint32_t)q, (uint32_t)d, r == 0, t - d * den == 0};
}

// floor(m * 5^i / 2^s): the product can reach 2^133, kept as hi * 2^64 + lo
inline Scaled scale_down(uint32_t m, int i, int s) {
    const u128 p = kT.pow5[i];
    const u128 t0 = (u128)m * (uint64_t)p;
    const u128 hi = (u128)m * (uint64_t)(p >> 64) + (t0 >> 64);
    const uint64_t lo = (uint64_t)t0;
    u128 q, r;
    if (s == 0) { q = (hi << 64) | lo; r = 0; }
    else if (s < 64) { q = (hi << (64 - s)) | (lo >> s); r = lo & (((uint64_t)1 << s) - 1); }
    else { const int h = s - 64; q = hi >> h; r = ((hi & (((u128)1 << h) - 1)) << 64) | lo; }
    const u128 t = r * 10;                                            // r < 2^105
    const u128 mask = s == 0 ? (u128)0 : (((u128)1 << s) - 1);
    return {(uint32_t)q, (uint32_t)(s == 0 ? 0 : (t >> s)), r == 0, (t & mask) == 0};
}

// shortest digits of a positive finite fp32: value ~ digits * 10^exp10, digits without trailing zeros
inline void shortest_f32(uint32_t bits, uint32_t* digits, int* exp10) {
    const uint32_t mant = bits & 0x7fffffu, ex = (bits >> 23) & 0xffu;
    uint32_t m2; int e2;
    if (ex == 0) { m2 = mant; e2 = 1 - 127 - 23 - 2; }
    else { m2 = mant | (1u << 23); e2 = (int)ex - 127 - 23 - 2; }
    const bool accept = (m2 & 1u) == 0;                               // round-half-even: the bounds read back as x
    const uint32_t mv = 4 * m2, mp = 4 * m2 + 2;
    const uint32_t mm = 4 * m2 - 1 - ((mant != 0 || ex <= 1) ? 1u : 0u);   // the gap below a power of two is half
    Scaled sv, sp, sm; int e10;
    if (e2 >= 0) {
        const int k = (e2 * 78913) >> 18;                             // floor(e2 * log10(2)), e2 <= 102
        sv = scale_up(mv, e2, k); sp = scale_up(mp, e2, k); sm = scale_up(mm, e2, k);
        e10 = k;
    } else {
        const int ne = -e2, i = ((ne * 78913) >> 18) + 1;             // ceil(ne * log10(2)): 10^i > 2^ne
        sv = scale_down(mv, i, ne - i); sp = scale_down(mp, i, ne - i); sm = scale_down(mm, i, ne - i);
        e10 = -i;
    }
    uint32_t vr = sv.q, vp = sp.q, vm = sm.q, last = sv.first;
    if (sp.exact && !accept) --vp;                                    // the upper bound itself is excluded
    bool vm_tz = sm.exact, vr_tz = sv.below_first_zero;
    int removed = 0;
    while (vp / 10 > vm / 10) {
        vm_tz &= vm % 10 == 0; vr_tz &= last == 0;
        last = vr % 10; vr /= 10; vp /= 10; vm /= 10; ++removed;
    }
    if (vm_tz && accept) {
        while (vm % 10 == 0) {
            vr_tz &= last == 0;
            last = vr % 10; vr /= 10; vp /= 10; vm /= 10; ++removed;
        }
    }
    if (vr_tz && last == 5 && vr % 2 == 0) last = 4;                  // exact tie: to the even digit
    uint32_t out = vr + (((vr == vm && (!accept || !vm_tz)) || last >= 5) ? 1u : 0u);
    int e = e10 + removed;
    while (out % 10 == 0) { out /= 10; ++e; }
    *digits = out; *exp10 = e;
}

inline char* put(char* p, const char* s) { while (*s) *p++ = *s++; return p; }

// str(np.float32(x)); at most 24 characters, returns one past the last written
inline char* format_f32(float x, char* p) {
    uint32_t bits; memcpy(&bits, &x, 4);
    if ((bits & 0x7fffffffu) > 0x7f800000u) return put(p, "nan");     // numpy prints nan without a sign
    if (bits >> 31) { *p++ = '-'; bits &= 0x7fffffffu; }
    if (bits == 0) return put(p, "0.0");
    if (bits == 0x7f800000u) return put(p, "inf");
    uint32_t d; int e;
    shortest_f32(bits, &d, &e);
    char dig[12]; int nd = 0;
    for (uint32_t t = d; t; t /= 10) dig[nd++] = (char)('0' + t % 10);   // reversed
    const int sci = e + nd - 1;                                       // exponent of the leading digit
    float ax; memcpy(&ax, &bits, 4);
    const double a = (double)ax;
    if (a < 1e-4 || a >= 1e16) {
        *p++ = dig[nd - 1];
        if (nd > 1) { *p++ = '.'; for (int i = nd - 2; i >= 0; --i) *p++ = dig[i]; }
        *p++ = 'e';
        int ae = sci;
        if (ae < 0) { *p++ = '-'; ae = -ae; } else *p++ = '+';
        *p++ = (char)('0' + ae / 10); *p++ = (char)('0' + ae % 10);   // |exponent| <= 45
        return p;
    }
    if (sci < 0) {
        *p++ = '0'; *p++ = '.';
        for (int i = -1; i > sci; --i) *p++ = '0';
        for (int i = nd - 1; i >= 0; --i) *p++ = dig[i];
        return p;
    }
    int i = nd - 1;
    for (int pos = sci; pos >= 0; --pos) *p++ = i >= 0 ? dig[i--] : '0';
    *p++ = '.';
    if (i < 0) *p++ = '0';
    while (i >= 0) *p++ = dig[i--];
    return p;
}

constexpr int kMaxF32Chars = 24;

}  // namespace

extern "C" {

int nbd_format_f32(float x, char* out24) {
    if (!out24) return -1;
    return (int)(format_f32(x, out24) - out24);
}

int nbd_format_f32_array(const float* x, int64_t n, char* out, int slot) {
    if (n < 0 || slot < kMaxF32Chars || (n > 0 && (!x || !out))) return NBD_E_BADARG;
    for (int64_t i = 0; i < n; ++i) {
        char* p = out + i * (int64_t)slot;
        char* e = format_f32(x[i], p);
        memset(e, 0, (size_t)(p + slot - e));
    }
    return 0;
}

size_t nbd_csv_state_bound(int n, size_t prefix_len, size_t mass_chars, size_t suffix_len) {
    if (n < 0) return 0;
    return (size_t)n * (prefix_len + suffix_len + 9 * (kMaxF32Chars + 1) + 1) + mass_chars;
}

int64_t nbd_csv_format_state(char* out, size_t cap, const char* prefix, size_t prefix_len, const char* mass_chars,
                             const int32_t* mass_off, const float* pos, const float* vel, const float* acc, int n,
                             const char* suffix, size_t suffix_len) {
    if (n < 0 || (n > 0 && (!out || !mass_chars || !mass_off || !pos || !vel || !acc))) return -1;
    if ((prefix_len && !prefix) || (suffix_len && !suffix)) return -1;
    if (n > 0 && cap < nbd_csv_state_bound(n, prefix_len, (size_t)(mass_off[n] - mass_off[0]), suffix_len)) return -1;
    char* p = out;
    const float* cols[3] = {pos, vel, acc};
    for (int i = 0; i < n; ++i) {
        memcpy(p, prefix, prefix_len); p += prefix_len;
        const int32_t ml = mass_off[i + 1] - mass_off[i];
        memcpy(p, mass_chars + mass_off[i], (size_t)ml); p += ml;
        for (int c = 0; c < 3; ++c)
            for (int k = 0; k < 3; ++k) { *p++ = ','; p = format_f32(cols[c][3 * (size_t)i + k], p); }
        memcpy(p, suffix, suffix_len); p += suffix_len;
    }
    return (int64_t)(p - out);
}

}  // extern "C"
