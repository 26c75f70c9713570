/*
 * pgas_canon.h -- constants of the canonical arithmetic shared by the HIP
 * kernels (…_amd/csrc) and the canonical C oracle (oracle/pgas_canon.c).
 *
 * Only constants and the RNG addressing scheme live here; the algorithm is written twice,
 * independently, on each side.
 */
#ifndef PGAS_CANON_H
#define PGAS_CANON_H

#include "pgas_detmath.h"

/* Particles are grouped into segments of PGAS_SEG consecutive indices.  Inside a segment the
 * softmax numerators exp(lw - m_seg) are quantised to PGAS_FIX_BITS fractional bits and
 * accumulated as 64-bit integers (exact, order independent); across segments the scaled segment
 * totals are combined in fp64 in the fixed "KS64 tree" order. */
#define PGAS_SEG 1024
#define PGAS_FIX_BITS 51
#define PGAS_FIX_SCALE 0x1p51
#define PGAS_FIX_INV 0x1p-51

/* limits of the compiled kernels / oracle */
#define PGAS_MAX_NX 4
#define PGAS_MAX_NY 2
#define PGAS_MAX_NU 4
#define PGAS_MAX_D 3
#define PGAS_MAX_J 64 /* distinct frequencies per basis dimension */

/* ---- softmax references.  The CDF of a weight vector is built on three levels, each with a
 * POWER-OF-TWO reference so that every rescaling between levels is exact:
 *   segment b (PGAS_SEG particles): kref_b = ceil(max_i lw_i * log2 e) (an integer-valued double, -inf for an empty segment),
 *       e_i = exp(lw_i - kref_b ln 2) in (0, 1] with the largest in (1/2, 1], quantised to 2^-51 and summed as integers;
 *   group g (PGAS_GRP consecutive segments): KG_g = max kref_b, segment totals rescaled by 2^(kref_b - KG_g);
 *   top (all groups): K = max KG_g, group totals rescaled by 2^(KG_g - K).
 * A contribution more than PGAS_LVL_FLUSH binary orders below the reference of the next level is exactly zero: every value
 * that is kept stays a normal number after both rescalings, which is what makes the hierarchy exact (no transcendental and
 * no rounding between levels; a workgroup only ever needs the group records plus the segments of the groups it searches). */
#define PGAS_GRP 64
#define PGAS_LVL_FLUSH 480.0
#define PGAS_SEG_LOG2E 0x1.71547652b82fep+0
#define PGAS_SEG_LN2_HI 0x1.62e42fefa39efp-1
#define PGAS_SEG_LN2_LO 0x1.abc9e3b39803fp-56
PGAS_HD double pgas_seg_ref(double m) { return __builtin_ceil(m * PGAS_SEG_LOG2E); }
PGAS_HD double pgas_seg_arg(double lw, double kref) {
    return PGAS_FMA(-kref, PGAS_SEG_LN2_LO, PGAS_FMA(-kref, PGAS_SEG_LN2_HI, lw));
}
/* 2^(k - K) for a reference k of one level under the reference K >= k of the next; 0 below the flush threshold, for empty
 * members (k = -inf) and when everything is empty (K = -inf: k - K is NaN) */
PGAS_HD double pgas_lvl_scale(double k, double K) {
    const double d = k - K; /* integer-valued and <= 0, -inf or NaN for empty members */
    return (d >= -PGAS_LVL_FLUSH) ? ldexp(1.0, (int)d) : 0.0;
}

/* Philox counter layout: (c0, c1, c2, c3) = (particle lo32, particle hi32, time step, stream | draw<<8),
 * key = (seed lo32, seed hi32). */
#define PGAS_STREAM_INIT 1u     /* x_0 ~ N(m0, P0)          (src/PGAS.py:167-172) */
#define PGAS_STREAM_PROP 2u     /* x_t ~ N(A phi, S)        (src/PGAS.py:72-75)   */
#define PGAS_STREAM_RESAMPLE 3u /* u of systematic_SISR     (src/Filtering.py:19)  */
#define PGAS_STREAM_ANCESTOR 4u /* u of the ancestor draw   (src/PGAS.py:123)      */
#define PGAS_STREAM_FINAL 5u    /* u of the final index     (src/PGAS.py:225)      */

PGAS_HD pgas_u32x4 pgas_rng_block(uint64_t seed, uint32_t stream, uint32_t draw, uint32_t t,
                                  uint64_t particle) {
    return pgas_philox4x32_10((uint32_t)particle, (uint32_t)(particle >> 32), t,
                              stream | (draw << 8), (uint32_t)seed, (uint32_t)(seed >> 32));
}

/* scalar uniform in (0,1) for (stream, t) */
PGAS_HD double pgas_rng_uniform(uint64_t seed, uint32_t stream, uint32_t t) {
    pgas_u32x4 w = pgas_rng_block(seed, stream, 0u, t, 0ull);
    return pgas_u52(w.v[0], w.v[1]);
}

/* normals z[0..n) of particle p at time t: draw d yields z[2d], z[2d+1] */
PGAS_HD void pgas_rng_normals(uint64_t seed, uint32_t stream, uint32_t t, uint64_t particle, int n,
                              double* z) {
    for (int d = 0; 2 * d < n; ++d) {
        double a, b;
        pgas_normal_pair(pgas_rng_block(seed, stream, (uint32_t)d, t, particle), &a, &b);
        z[2 * d] = a;
        if (2 * d + 1 < n) z[2 * d + 1] = b;
    }
}

/* ---- streams of the marginalised family (reference src/Algorithm1.py, src/Algorithm3.py); int-var index i < 8 is added
 * to the *_INTVAR streams */
#define PGAS_STREAM_M_INIT_STATE 16u  /* x_0 ~ N(m0, P0)                      (src/Algorithm1.py:139-145) */
#define PGAS_STREAM_M_STATE 17u       /* process noise of SSM.draw_state      (src/StateSpaceModel.py:67-74) */
#define PGAS_STREAM_M_RESAMPLE 18u    /* u of systematic_SISR                 (src/Algorithm1.py:346-347) */
#define PGAS_STREAM_M_ANCESTOR 19u    /* u of the reference ancestor draw     (src/Algorithm3.py:119-122) */
#define PGAS_STREAM_M_FINAL 20u       /* u of the final index                 (src/Algorithm3.py:287) */
#define PGAS_STREAM_M_INIT_INTVAR 24u /* xi_0 ~ N(mean, cov)                  (src/Algorithm1.py:146-153) */
#define PGAS_STREAM_M_INTVAR 32u      /* Student-t of prior_mniw_drawPred     (src/BayesianInferrence.py:104) */

/* Gamma(a, 1) variate of particle p at (stream, t) by Marsaglia-Tsang squeeze-free rejection; attempt k consumes Philox
 * draws 1 + 2k (candidate normal) and 2 + 2k (acceptance uniform, boost uniform); a < 1 uses G(a) = G(a+1) U^(1/a).
 * Built from the deterministic primitives only, so host and device agree bit for bit. */
PGAS_HD double pgas_rng_gamma(uint64_t seed, uint32_t stream, uint32_t t, uint64_t particle, double a) {
    const double a1 = a < 1.0 ? a + 1.0 : a;
    const double d = a1 - 1.0 / 3.0;
    const double c = 1.0 / sqrt(9.0 * d);
    double g = d;
    double boost = 1.0;
    for (uint32_t k = 0; k < 32u; ++k) {
        double x, spare;
        pgas_normal_pair(pgas_rng_block(seed, stream, 1u + 2u * k, t, particle), &x, &spare);
        const pgas_u32x4 w = pgas_rng_block(seed, stream, 2u + 2u * k, t, particle);
        const double u = pgas_u52(w.v[0], w.v[1]);
        if (k == 0u && a < 1.0) boost = pgas_exp(pgas_log(pgas_u52(w.v[2], w.v[3])) / a);
        double v = PGAS_FMA(c, x, 1.0);
        if (v <= 0.0) continue;
        v = v * v * v;
        g = d * v;
        if (pgas_log(u) < PGAS_FMA(0.5 * x, x, d) - g + d * pgas_log(v)) break;
    }
    return g * boost;
}

/* Student-t(nu) variate: z sqrt((nu/2) / G), G ~ Gamma(nu/2, 1) (= z / sqrt(chi2_nu / nu)); z is the first normal of draw 0 */
PGAS_HD double pgas_rng_student_t(uint64_t seed, uint32_t stream, uint32_t t, uint64_t particle, double nu) {
    double z, spare;
    pgas_normal_pair(pgas_rng_block(seed, stream, 0u, t, particle), &z, &spare);
    const double a = 0.5 * nu;
    return z * sqrt(a / pgas_rng_gamma(seed, stream, t, particle, a));
}

#endif /* PGAS_CANON_H */
