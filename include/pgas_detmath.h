/*
 * pgas_detmath.h -- deterministic fp64 primitives shared by the HIP kernels and the
 * canonical C oracle.
 *
 * Why this exists: the conditional-SMC sweep (reference src/PGAS.py:79-153) turns fp64 weights
 * into integer ancestor indices (src/Filtering.py:28-35). One ulp of difference between a host
 * libm exp() and the device exp() flips an index at N = 2^20, and with the reference's
 * un-resampled propagation (quirk Q1) that flip changes every later weight.  To make "GPU engine
 * == CPU oracle" a bit-exact statement at any N, both sides evaluate exp / log / sin(pi x) /
 * cos(pi x) / the counter-based RNG with THIS header: plain IEEE-754 double +,-,*,/,sqrt,fma and
 * integer arithmetic only, no libm, no vendor math library.  Every operation below is correctly
 * rounded on x86-64 (SSE2 + FMA3) and on gfx950 (v_fma_f64, v_rndne_f64, IEEE v_div/v_sqrt
 * expansions), so the results are identical bit for bit provided the translation unit is built
 * with -ffp-contract=off (the only fused operations are the explicit PGAS_FMA calls).
 *
 * Accuracy (tests/test_detmath.py, against mpmath): exp < 1 ulp, log < 1 ulp on (0,1],
 * sinpi/cospi <= 1 ulp on the reduced interval.  Coefficients: tools/gen_detmath_coeffs.py.
 *
 * This file is part of the specification ("canonical arithmetic"), not of
 * the oracle: the oracle (oracle/pgas_canon.c) is a separately written serial implementation of
 * the algorithm that calls these primitives.
 */
#ifndef PGAS_DETMATH_H
#define PGAS_DETMATH_H

#include <stdint.h>

#if defined(__HIPCC__) || defined(__HIP__)
#include <hip/hip_runtime.h>
#define PGAS_HD __host__ __device__ __forceinline__
#else
#define PGAS_HD static inline
#endif

#define PGAS_FMA(a, b, c) __builtin_fma((a), (b), (c))

typedef union {
    double d;
    uint64_t u;
} pgas_du_t;

PGAS_HD uint64_t pgas_d2bits(double x) {
    pgas_du_t t;
    t.d = x;
    return t.u;
}
PGAS_HD double pgas_bits2d(uint64_t u) {
    pgas_du_t t;
    t.u = u;
    return t.d;
}

/* ---------------------------------------------------------------- integer <-> double ---- */

/* exact for any c; one rounding (the final add), round-to-nearest-even on both targets */
PGAS_HD double pgas_u64_to_double(uint64_t c) {
    return (double)(uint32_t)(c >> 32) * 4294967296.0 + (double)(uint32_t)c;
}

/* v must be integer valued, 0 <= v < 2^63 */
PGAS_HD uint64_t pgas_double_to_u64(double v) {
    double hi = __builtin_floor(v * (1.0 / 4294967296.0));
    double lo = v - hi * 4294967296.0; /* exact */
    return ((uint64_t)(uint32_t)hi << 32) | (uint64_t)(uint32_t)lo;
}

/* ------------------------------------------------------------------------------ exp ---- */

/* Batch forms (pgas_*_n): the same arithmetic applied to n <= PGAS_NB independent arguments, written
 * coefficient-major so that device code keeps each polynomial coefficient in registers for the whole
 * batch (a scalar Horner chain costs two extra register moves per fp64 literal on gfx950).  The scalar
 * functions are the n = 1 instances, so there is exactly one definition of every operation sequence. */
#define PGAS_NB 8

/* exp(x).  Defined as 0 for x < -708 (no subnormal results), +inf for x > 709, NaN -> NaN. */
PGAS_HD void pgas_exp_n(const double* x, double* out, int n) {
    const double LOG2E = 0x1.71547652b82fep+0;
    const double LN2_HI = 0x1.62e42fefa39efp-1;
    const double LN2_LO = 0x1.abc9e3b39803fp-56;
    const double C[13] = {0x1.1eed8eff8d898p-29, 0x1.ae64567f544e4p-26, 0x1.27e4fb7789f5cp-22, 0x1.71de3a556c734p-19,
                          0x1.a01a01a01a01ap-16, 0x1.a01a01a01a01ap-13, 0x1.6c16c16c16c17p-10, 0x1.1111111111111p-7,
                          0x1.5555555555555p-5,  0x1.5555555555555p-3,  0.5, 1.0, 1.0};
    double k[PGAS_NB], r[PGAS_NB], p[PGAS_NB];
    for (int i = 0; i < n; ++i) {
        /* clamp the argument of the reduction so that k stays in range; the final select restores the edge cases */
        double xc = x[i] > 709.0 ? 709.0 : (x[i] < -708.0 ? -708.0 : x[i]);
        k[i] = __builtin_rint(xc * LOG2E);
        r[i] = PGAS_FMA(-k[i], LN2_HI, xc);
        r[i] = PGAS_FMA(-k[i], LN2_LO, r[i]);
        p[i] = 0x1.6124613a86d09p-33; /* 1/13! */
    }
    for (int j = 0; j < 13; ++j)
        for (int i = 0; i < n; ++i) p[i] = PGAS_FMA(p[i], r[i], C[j]);
    for (int i = 0; i < n; ++i) {
        /* p in (0.70, 1.42); k in [-1021, 1023]: the scaled result is a normal number */
        int64_t ki = (int64_t)(int32_t)k[i];
        double v = pgas_bits2d(pgas_d2bits(p[i]) + ((uint64_t)ki << 52));
        if (x[i] > 709.0) v = __builtin_inf();
        if (x[i] < -708.0) v = 0.0;
        if (x[i] != x[i]) v = x[i];
        out[i] = v;
    }
}
PGAS_HD double pgas_exp(double x) {
    double o;
    pgas_exp_n(&x, &o, 1);
    return o;
}

/* ------------------------------------------------------------------------------ log ---- */

/* log(x) for positive normal x (the RNG only calls it on (0,1)).  x <= 0 or NaN -> NaN. */
PGAS_HD void pgas_log_n(const double* x, double* out, int n) {
    const double LN2_HI = 0x1.62e42fee00000p-1; /* 21 trailing zero bits: k*LN2_HI exact */
    const double LN2_LO = 0x1.a39ef35793c76p-33;
    const double C[11] = {0x1.642c8590b2164p-4, 0x1.8618618618618p-4, 0x1.af286bca1af28p-4, 0x1.e1e1e1e1e1e1ep-4,
                          0x1.1111111111111p-3, 0x1.3b13b13b13b14p-3, 0x1.745d1745d1746p-3, 0x1.c71c71c71c71cp-3,
                          0x1.2492492492492p-2, 0x1.999999999999ap-2, 0x1.5555555555555p-1};
    double f[PGAS_NB], k[PGAS_NB], s[PGAS_NB], z[PGAS_NB], R[PGAS_NB];
    for (int i = 0; i < n; ++i) {
        uint64_t b = pgas_d2bits(x[i]);
        int32_t e = (int32_t)(b >> 52) - 1023;
        uint64_t mant = b & 0x000fffffffffffffULL;
        /* m in [1,2); fold to [sqrt(1/2), sqrt(2)) */
        int big = mant > 0x6a09e667f3bcdULL; /* m > sqrt(2) */
        e += big;
        b = mant | (big ? 0x3fe0000000000000ULL : 0x3ff0000000000000ULL);
        f[i] = pgas_bits2d(b) - 1.0; /* exact */
        k[i] = (double)e;
        s[i] = f[i] / (2.0 + f[i]);
        z[i] = s[i] * s[i];
        R[i] = 0x1.47ae147ae147bp-4; /* 2/25 */
    }
    for (int j = 0; j < 11; ++j)
        for (int i = 0; i < n; ++i) R[i] = PGAS_FMA(R[i], z[i], C[j]);
    for (int i = 0; i < n; ++i) {
        double Rz = R[i] * z[i];
        double hfsq = 0.5 * f[i] * f[i];
        /* log(1+f) = f - (hfsq - s*(hfsq+R)) */
        double t = PGAS_FMA(s[i], hfsq + Rz, k[i] * LN2_LO);
        double v = k[i] * LN2_HI - ((hfsq - t) - f[i]);
        if (!(x[i] > 0.0)) v = __builtin_nan("");
        out[i] = v;
    }
}
PGAS_HD double pgas_log(double x) {
    double o;
    pgas_log_n(&x, &o, 1);
    return o;
}

/* ------------------------------------------------------------------- sin/cos(pi r) ---- */

/* (sin(pi r), cos(pi r)).  Exact zeros at integers / half-integers.  |r| >= 2^30 -> (0, 1). */
PGAS_HD void pgas_sincospi_n(const double* r, double* sp, double* cp, int n) {
    const double PI_HI = 0x1.921fb54442d18p+1;
    const double PI_LO = 0x1.1a62633145c07p-53;
    const double S[8] = {0x1.aaec32af93359p-21, -0x1.6fadb9f155744p-16, 0x1.e8f434d018d63p-12, -0x1.e3074fde8871fp-8,
                         0x1.50783487ee782p-4,  -0x1.32d2cce62bd86p-1,  0x1.466bc6775aae2p+1,  -0x1.4abbce625be53p+2};
    const double Cc[8] = {0x1.20c62c2f2d7f5p-18, -0x1.b6e24f44b128fp-14, 0x1.f9d38a3763cc3p-10, -0x1.a6d1f2a204a8cp-6,
                          0x1.e1f506891babbp-3,  -0x1.55d3c7e3cbffap+0,  0x1.03c1f081b5ac4p+2,  -0x1.3bd3cc9be45dep+2};
    double nn[PGAS_NB], f[PGAS_NB], z[PGAS_NB], ps[PGAS_NB], pc[PGAS_NB];
    for (int i = 0; i < n; ++i) {
        int ok = __builtin_fabs(r[i]) < 0x1p30; /* false for NaN too */
        double rr = ok ? r[i] : 0.0;
        nn[i] = __builtin_rint(rr + rr);      /* nearest half-integer index, |nn| < 2^31 */
        f[i] = PGAS_FMA(-0.5, nn[i], rr);     /* exact, |f| <= 1/4 */
        z[i] = f[i] * f[i];
        ps[i] = -0x1.8a404211f9547p-26;
        pc[i] = -0x1.2a0c591af8314p-23;
    }
    for (int j = 0; j < 8; ++j)
        for (int i = 0; i < n; ++i) {
            ps[i] = PGAS_FMA(ps[i], z[i], S[j]);
            pc[i] = PGAS_FMA(pc[i], z[i], Cc[j]);
        }
    for (int i = 0; i < n; ++i) {
        double fz = f[i] * z[i];
        double s = PGAS_FMA(f[i], PI_HI, PGAS_FMA(fz, ps[i], f[i] * PI_LO));
        double c = PGAS_FMA(z[i], pc[i], 1.0);
        /* quadrant = nn mod 4: odd -> swap; sin negative in quadrants 2,3; cos negative in 1,2 */
        uint32_t q = (uint32_t)(int32_t)nn[i];
        uint64_t sb = pgas_d2bits((q & 1u) ? c : s);
        uint64_t cb = pgas_d2bits((q & 1u) ? s : c);
        sb ^= (uint64_t)(q & 2u) << 62;
        cb ^= (uint64_t)((q + 1u) & 2u) << 62;
        double so = pgas_bits2d(sb), co = pgas_bits2d(cb);
        if (r[i] != r[i]) {
            so = r[i];
            co = r[i];
        }
        sp[i] = so;
        cp[i] = co;
    }
}
PGAS_HD void pgas_sincospi(double r, double* sp, double* cp) { pgas_sincospi_n(&r, sp, cp, 1); }

/* --------------------------------------------------------------------- Philox4x32-10 ---- */
/* Salmon, Moraes, Dror, Shaw, "Parallel random numbers: as easy as 1, 2, 3" (SC'11).
 * Known-answer vectors from the Random123 distribution are checked in tests/test_detmath.py. */

typedef struct {
    uint32_t v[4];
} pgas_u32x4;

PGAS_HD pgas_u32x4 pgas_philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                      uint32_t k0, uint32_t k1) {
    for (int i = 0; i < 10; ++i) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0;
        c1 = n1;
        c2 = n2;
        c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    pgas_u32x4 r;
    r.v[0] = c0;
    r.v[1] = c1;
    r.v[2] = c2;
    r.v[3] = c3;
    return r;
}

/* 52-bit uniform strictly inside (0,1): (n + 1/2) * 2^-52, n = top 52 bits of (hi:lo).
 * n + 1/2 < 2^52 is exactly representable, so min = 2^-53 and max = 1 - 2^-53. */
PGAS_HD double pgas_u52(uint32_t lo, uint32_t hi) {
    uint64_t n = (((uint64_t)hi << 32) | lo) >> 12;
    return ((double)(uint32_t)(n >> 32) * 4294967296.0 + (double)(uint32_t)n + 0.5) * 0x1p-52;
}

/* Two independent standard normals from each Philox block (Box-Muller), n <= PGAS_NB blocks at once. */
PGAS_HD void pgas_normal_pair_n(const pgas_u32x4* w, double* z0, double* z1, int n) {
    double ua[PGAS_NB], ub2[PGAS_NB], lg[PGAS_NB], s[PGAS_NB], c[PGAS_NB];
    for (int i = 0; i < n; ++i) {
        ua[i] = pgas_u52(w[i].v[0], w[i].v[1]);
        double ub = pgas_u52(w[i].v[2], w[i].v[3]);
        ub2[i] = ub + ub;
    }
    pgas_log_n(ua, lg, n);
    pgas_sincospi_n(ub2, s, c, n);
    for (int i = 0; i < n; ++i) {
        double rad = __builtin_sqrt(-2.0 * lg[i]);
        z0[i] = rad * c[i];
        z1[i] = rad * s[i];
    }
}
PGAS_HD void pgas_normal_pair(pgas_u32x4 w, double* z0, double* z1) { pgas_normal_pair_n(&w, z0, z1, 1); }

#endif /* PGAS_DETMATH_H */
