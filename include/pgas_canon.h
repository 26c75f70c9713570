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

#endif /* PGAS_CANON_H */
