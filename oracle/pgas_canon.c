/*
 * pgas_canon.c -- canonical-arithmetic CPU oracle for the conditional-SMC sweep.
 *
 * TEST INFRASTRUCTURE ONLY: loaded by tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg.  The product package never links or loads it.
 *
 * PARITY UNPINNED against the JAX reference (it cannot be imported offline and ships no golden
 * vectors).  This file is pinned (a) against oracle/pgas_numpy.py, the literal
 * NumPy restatement of the reference, to 1e-12 on every continuous quantity and index-for-index
 * away from CDF ties (tests/test_oracle_canon.py), and (b) by the analytic KATs of SURVEY 8c.
 *
 * What it is: a plain serial C restatement of reference src/PGAS.py:45-228 and
 * src/Filtering.py:6-55 in the "canonical arithmetic" of DESIGN.md section 4 -- the same
 * algorithm, with every floating-point reduction given a fixed order and every transcendental
 * taken from include/pgas_detmath.h, so that the HIP engine can be required to match it BIT FOR
 * BIT at any particle count.  It is written independently of the HIP kernels (serial loops, no
 * tiling, no shared code beyond the arithmetic primitives and constants in include/).
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -mfma).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/pgas_canon.h"

#define EXPORT __attribute__((visibility("default")))

typedef struct {
    int32_t N, T, nx, ny, nu, D, M;
    int32_t J[PGAS_MAX_D], j0[PGAS_MAX_D], jstep[PGAS_MAX_D], sel[PGAS_MAX_D];
    double alpha[PGAS_MAX_D], beta[PGAS_MAX_D];
    double nrm;
    double H[PGAS_MAX_NY * PGAS_MAX_NX];
    double LRinv[PGAS_MAX_NY * PGAS_MAX_NY];
    double cR;
    int32_t* idx; /* M x D frequencies (reference order, src/BasisFunctions.py:59) */
    double* y;    /* T x ny */
    double* u;    /* T x nu */
    int32_t corrected; /* 1: propagate from the resampled ancestors (NOT the reference's behaviour, quirk Q1 removed) */
} oc_model;

/* --------------------------------------------------------------------------- test hooks -- */
EXPORT void oc_exp_v(const double* x, double* out, int64_t n) {
    for (int64_t i = 0; i < n; ++i) out[i] = pgas_exp(x[i]);
}
EXPORT void oc_log_v(const double* x, double* out, int64_t n) {
    for (int64_t i = 0; i < n; ++i) out[i] = pgas_log(x[i]);
}
EXPORT void oc_sincospi_v(const double* x, double* s, double* c, int64_t n) {
    for (int64_t i = 0; i < n; ++i) pgas_sincospi(x[i], &s[i], &c[i]);
}
EXPORT void oc_philox(const uint32_t* ctr, const uint32_t* key, uint32_t* out) {
    pgas_u32x4 r = pgas_philox4x32_10(ctr[0], ctr[1], ctr[2], ctr[3], key[0], key[1]);
    memcpy(out, r.v, sizeof r.v);
}
EXPORT double oc_uniform(uint64_t seed, uint32_t stream, uint32_t t) {
    return pgas_rng_uniform(seed, stream, t);
}
EXPORT void oc_normals(uint64_t seed, uint32_t stream, uint32_t t, int64_t p0, int64_t np, int n,
                       double* z) {
    for (int64_t p = 0; p < np; ++p) pgas_rng_normals(seed, stream, t, (uint64_t)(p0 + p), n, z + p * n);
}
/* Student-t(nu[p]) variates of particles p0 .. p0+np (marginalised family, BI:104) */
EXPORT void oc_chi2(uint64_t seed, uint32_t stream, uint32_t t, int64_t p0, int64_t np, const double* nu, double* out) {
    for (int64_t p = 0; p < np; ++p) out[p] = 2.0 * pgas_rng_gamma(seed, stream, t, (uint64_t)(p0 + p), 0.5 * nu[p]);
}

EXPORT void oc_student_t(uint64_t seed, uint32_t stream, uint32_t t, int64_t p0, int64_t np, const double* nu, double* out) {
    for (int64_t p = 0; p < np; ++p) out[p] = pgas_rng_student_t(seed, stream, t, (uint64_t)(p0 + p), nu[p]);
}
EXPORT void oc_gamma(uint64_t seed, uint32_t stream, uint32_t t, int64_t p0, int64_t np, const double* a, double* out) {
    for (int64_t p = 0; p < np; ++p) out[p] = pgas_rng_gamma(seed, stream, t, (uint64_t)(p0 + p), a[p]);
}
EXPORT double oc_u64_to_double(uint64_t c) { return pgas_u64_to_double(c); }
EXPORT int32_t oc_seg(void) { return PGAS_SEG; }

/* ------------------------------------------------------------------------------- model -- */
EXPORT oc_model* oc_model_create(int32_t N, int32_t T, int32_t nx, int32_t ny, int32_t nu, int32_t D,
                                 int32_t M, const int32_t* idx, const int32_t* sel,
                                 const double* alpha, const double* beta, double nrm,
                                 const double* H, const double* LRinv, double cR, const double* y,
                                 const double* u) {
    if (nx < 1 || nx > PGAS_MAX_NX || ny < 1 || ny > PGAS_MAX_NY || nu < 0 || nu > PGAS_MAX_NU ||
        D < 1 || D > PGAS_MAX_D || M < 1 || N < 1 || T < 1)
        return NULL;
    oc_model* m = (oc_model*)calloc(1, sizeof *m);
    m->N = N; m->T = T; m->nx = nx; m->ny = ny; m->nu = nu; m->D = D; m->M = M;
    m->nrm = nrm; m->cR = cR;
    m->idx = (int32_t*)malloc(sizeof(int32_t) * (size_t)M * D);
    memcpy(m->idx, idx, sizeof(int32_t) * (size_t)M * D);
    for (int d = 0; d < D; ++d) {
        m->sel[d] = sel[d]; m->alpha[d] = alpha[d]; m->beta[d] = beta[d];
        /* per-dimension frequency progression j0, j0+step, ... (src/BasisFunctions.py:24-25) */
        int lo = idx[d], hi = idx[d], second = 0;
        for (int k = 0; k < M; ++k) {
            int j = idx[k * D + d];
            if (j < lo) lo = j;
            if (j > hi) hi = j;
        }
        for (int k = 0; k < M; ++k) {
            int j = idx[k * D + d];
            if (j > lo && (second == 0 || j < second)) second = j;
        }
        m->j0[d] = lo;
        m->jstep[d] = second ? second - lo : 1;
        m->J[d] = (hi - lo) / m->jstep[d] + 1;
        if (m->J[d] > PGAS_MAX_J) { free(m->idx); free(m); return NULL; }
    }
    memcpy(m->H, H, sizeof(double) * ny * nx);
    memcpy(m->LRinv, LRinv, sizeof(double) * ny * ny);
    m->y = (double*)malloc(sizeof(double) * (size_t)T * ny);
    memcpy(m->y, y, sizeof(double) * (size_t)T * ny);
    m->u = (double*)malloc(sizeof(double) * ((size_t)T * nu + 1));
    if (nu) memcpy(m->u, u, sizeof(double) * (size_t)T * nu);
    return m;
}
EXPORT void oc_model_destroy(oc_model* m) {
    if (!m) return;
    free(m->idx); free(m->y); free(m->u); free(m);
}
EXPORT void oc_model_grid(const oc_model* m, int32_t* J, int32_t* j0, int32_t* jstep) {
    for (int d = 0; d < m->D; ++d) { J[d] = m->J[d]; j0[d] = m->j0[d]; jstep[d] = m->jstep[d]; }
}

static int64_t grid_size(const oc_model* m) {
    int64_t g = 1;
    for (int d = 0; d < m->D; ++d) g *= m->J[d];
    return g;
}
static int64_t grid_pos(const oc_model* m, int k) {
    int64_t pos = 0;
    for (int d = 0; d < m->D; ++d) pos = pos * m->J[d] + (m->idx[k * m->D + d] - m->j0[d]) / m->jstep[d];
    return pos;
}

/* G[k][grid] = A[k][m] * nrm on the dense frequency grid, zero where no basis function sits */
EXPORT void oc_pack_coeff(const oc_model* m, const double* A, double* G) {
    int64_t g = grid_size(m);
    memset(G, 0, sizeof(double) * (size_t)(m->nx * g));
    for (int k = 0; k < m->nx; ++k)
        for (int b = 0; b < m->M; ++b) G[k * g + grid_pos(m, b)] = A[k * m->M + b] * m->nrm;
}
EXPORT int64_t oc_grid_size(const oc_model* m) { return grid_size(m); }

/* sin(pi * j * r_d) for the J[d] frequencies j = j0 + q step of dimension d  (src/BasisFunctions.py:77-80).
 * Canonical recurrences: one-dimensional bases rotate (sin, cos) by the step angle;
 * multi-dimensional bases use the Chebyshev three-term recurrence s_{q+1} = 2 cos(step) s_q - s_{q-1}. */
static void dim_sines(const oc_model* m, int d, const double* v, double* s) {
    double r = PGAS_FMA(v[m->sel[d]], m->alpha[d], m->beta[d]);
    double sc, cc, sd, cd;
    pgas_sincospi((double)m->j0[d] * r, &sc, &cc);
    if (m->jstep[d] == m->j0[d]) {
        sd = sc;
        cd = cc;
    } else {
        pgas_sincospi((double)m->jstep[d] * r, &sd, &cd);
    }
    if (m->D == 1) {
        for (int q = 0; q < m->J[d]; ++q) {
            s[q] = sc;
            double sn = PGAS_FMA(sc, cd, cc * sd);
            double cn = PGAS_FMA(cc, cd, -(sc * sd));
            sc = sn; cc = cn;
        }
    } else {
        double prev = (m->jstep[d] == m->j0[d]) ? 0.0 : PGAS_FMA(sc, cd, -(cc * sd)); /* sin(pi (j0 - step) r) */
        double cur = sc, tw = cd + cd;
        for (int q = 0; q < m->J[d]; ++q) {
            s[q] = cur;
            double nx = PGAS_FMA(tw, cur, -prev);
            prev = cur;
            cur = nx;
        }
    }
}

/* aux = A phi(x,u)  (src/PGAS.py:52-55) as a nested contraction over the frequency grid */
static void eval_mean(const oc_model* m, const double* G, const double* xp, const double* ut, double* aux) {
    double v[PGAS_MAX_NX + PGAS_MAX_NU];
    double s[PGAS_MAX_D][PGAS_MAX_J];
    for (int k = 0; k < m->nx; ++k) v[k] = xp[k];
    for (int k = 0; k < m->nu; ++k) v[m->nx + k] = ut[k];
    for (int d = 0; d < m->D; ++d) dim_sines(m, d, v, s[d]);
    int64_t g = grid_size(m);
    const int J0 = m->J[0], J1 = m->D > 1 ? m->J[1] : 1, J2 = m->D > 2 ? m->J[2] : 1;
    for (int k = 0; k < m->nx; ++k) {
        const double* Gk = G + k * g;
        double acc0 = 0.0;
        if (m->D == 1) {
            for (int a = 0; a < J0; ++a) acc0 = PGAS_FMA(Gk[a], s[0][a], acc0);
        } else if (m->D == 2) {
            for (int a = 0; a < J0; ++a) {
                double acc1 = 0.0;
                for (int b = 0; b < J1; ++b) acc1 = PGAS_FMA(Gk[a * J1 + b], s[1][b], acc1);
                acc0 = PGAS_FMA(s[0][a], acc1, acc0);
            }
        } else {
            for (int a = 0; a < J0; ++a) {
                double acc1 = 0.0;
                for (int b = 0; b < J1; ++b) {
                    double acc2 = 0.0;
                    for (int c = 0; c < J2; ++c) acc2 = PGAS_FMA(Gk[(a * J1 + b) * J2 + c], s[2][c], acc2);
                    acc1 = PGAS_FMA(s[1][b], acc2, acc1);
                }
                acc0 = PGAS_FMA(s[0][a], acc1, acc0);
            }
        }
        aux[k] = acc0;
    }
}

/* phi_m(x) in reference order -- test hook for pgas_basis_eval */
EXPORT void oc_basis_eval(const oc_model* m, const double* x, int32_t t, int64_t np, double* phi) {
    double v[PGAS_MAX_NX + PGAS_MAX_NU];
    double s[PGAS_MAX_D][PGAS_MAX_J];
    for (int64_t p = 0; p < np; ++p) {
        for (int k = 0; k < m->nx; ++k) v[k] = x[p * m->nx + k];
        for (int k = 0; k < m->nu; ++k) v[m->nx + k] = m->u[(size_t)t * m->nu + k];
        for (int d = 0; d < m->D; ++d) dim_sines(m, d, v, s[d]);
        for (int b = 0; b < m->M; ++b) {
            double f = m->nrm;
            for (int d = 0; d < m->D; ++d) f = f * s[d][(m->idx[b * m->D + d] - m->j0[d]) / m->jstep[d]];
            phi[p * m->M + b] = f;
        }
    }
}

/* log N(y; H x, R)  (likelihood_fcn of src/Toy_Example.py:142-144, src/EMPS.py:250-252) */
static double loglik(const oc_model* m, const double* yt, const double* xv) {
    double e[PGAS_MAX_NY], quad = 0.0;
    for (int j = 0; j < m->ny; ++j) {
        e[j] = yt[j];
        for (int k = 0; k < m->nx; ++k) e[j] = PGAS_FMA(-m->H[j * m->nx + k], xv[k], e[j]);
    }
    for (int j = 0; j < m->ny; ++j) {
        double w = 0.0;
        for (int l = 0; l <= j; ++l) w = PGAS_FMA(m->LRinv[j * m->ny + l], e[l], w);
        quad = PGAS_FMA(w, w, quad);
    }
    return PGAS_FMA(-0.5, quad, m->cR);
}

/* ----------------------------------------------------------------- per-segment softmax -- */
/* lw[n] -> segment max m, quantised inclusive cumsum c[n], total s  (canonical softmax numerators) */
static void segment_scan(const double* lw, int n, double* mo, uint64_t* c, uint64_t* so) {
    double mx = -INFINITY;
    for (int i = 0; i < n; ++i)
        if (lw[i] > mx) mx = lw[i];
    const double kref = pgas_seg_ref(mx); /* power-of-two reference, include/pgas_canon.h */
    uint64_t run = 0;
    for (int i = 0; i < n; ++i) {
        double e = pgas_exp(pgas_seg_arg(lw[i], kref));
        uint64_t q = (e > 0.0) ? pgas_double_to_u64(__builtin_rint(e * PGAS_FIX_SCALE)) : 0;
        run += q;
        c[i] = run;
    }
    *mo = kref;
    *so = run;
}

/* Canonical inclusive scan of one group of 64 (zero padded), the "R16 tree" of DESIGN.md 4.4:
 * Kogge-Stone inside each row of 16 (offsets 1,2,4,8), then rows 1 and 3 add the total of the row before them,
 * then rows 2 and 3 add the value of element 31.  Elements without a partner add 0.0. */
static void ks64(double* v) {
    double t[64];
    for (int off = 1; off < 16; off <<= 1) {
        for (int l = 0; l < 64; ++l) t[l] = v[l] + ((l % 16) >= off ? v[l - off] : 0.0);
        memcpy(v, t, sizeof t);
    }
    for (int l = 0; l < 64; ++l) t[l] = v[l] + (((l / 16) & 1) ? v[(l / 16) * 16 - 1] : 0.0);
    memcpy(v, t, sizeof t);
    for (int l = 0; l < 64; ++l) t[l] = v[l] + ((l / 32) ? v[31] : 0.0);
    memcpy(v, t, sizeof t);
}

/* ---- the hierarchical CDF of DESIGN.md 4.4: segment -> group of PGAS_GRP segments -> top.
 *   group g:  KG = max kref_b;  t_b = 2^(kref_b - KG) (s_b 2^-51);  e_b = exclusive R16 prefix of t inside the group (64 lanes,
 *             missing segments are zeros);  w_b = e_b + t_b;  m_b = running maximum of w inside the group;  TG = m of the
 *             group's last segment
 *   top:      K = max KG;  TT_g = 2^(KG_g - K) TG_g;  E_g = exclusive prefix of TT (R16 inside blocks of 64 groups, R16 over the
 *             block totals, E = EC + EB);  W_g = E_g + TT_g;  CM_g = running maximum of W;  S = CM_last
 *   particle k of segment b of group g:  v_k = max(m_{b-1}, e_b + sc_b (c_k 2^-51)),  num_k = max(CM_{g-1}, E_g + sig_g v_k)
 * num is non-decreasing in k by construction; every rescaling is by an exact power of two (include/pgas_canon.h). */
typedef struct {
    int nseg, n1;
    double *e, *sc, *m;      /* per segment */
    double *E, *sig, *CM;    /* per group */
    double S;
    int valid;
} upper_t;

static void upper_build(upper_t* U, int nseg, const double* segk, const uint64_t* segs) {
    const int n1 = (nseg + PGAS_GRP - 1) / PGAS_GRP, n2 = (n1 + 63) / 64;
    U->nseg = nseg;
    U->n1 = n1;
    U->e = (double*)malloc(sizeof(double) * nseg);
    U->sc = (double*)malloc(sizeof(double) * nseg);
    U->m = (double*)malloc(sizeof(double) * nseg);
    U->E = (double*)malloc(sizeof(double) * n1);
    U->sig = (double*)malloc(sizeof(double) * n1);
    U->CM = (double*)malloc(sizeof(double) * n1);
    double* KG = (double*)malloc(sizeof(double) * n1);
    double* TT = (double*)calloc((size_t)n2 * 64, sizeof(double));
    double K = -INFINITY;
    for (int g = 0; g < n1; ++g) {
        const int b0 = g * PGAS_GRP, nb = nseg - b0 < PGAS_GRP ? nseg - b0 : PGAS_GRP;
        double kg = -INFINITY, t[64], inc[64];
        for (int l = 0; l < nb; ++l)
            if (segk[b0 + l] > kg) kg = segk[b0 + l];
        memset(t, 0, sizeof t);
        for (int l = 0; l < nb; ++l) {
            U->sc[b0 + l] = pgas_lvl_scale(segk[b0 + l], kg);
            t[l] = U->sc[b0 + l] * (pgas_u64_to_double(segs[b0 + l]) * PGAS_FIX_INV);
        }
        memcpy(inc, t, sizeof t);
        ks64(inc);
        double run = 0.0;
        for (int l = 0; l < nb; ++l) { /* TG = m of the last REAL segment of the group: padding lanes take no part */
            const double e = l ? inc[l - 1] : 0.0, w = e + t[l];
            if (w > run) run = w;
            U->e[b0 + l] = e;
            U->m[b0 + l] = run;
        }
        KG[g] = kg;
        TT[g] = run; /* TG for now */
        if (kg > K) K = kg;
    }
    for (int g = 0; g < n1; ++g) {
        U->sig[g] = pgas_lvl_scale(KG[g], K);
        TT[g] = U->sig[g] * TT[g];
    }
    double* incB = (double*)malloc(sizeof(double) * (size_t)n2 * 64);
    double incC[64];
    memcpy(incB, TT, sizeof(double) * (size_t)n2 * 64);
    memset(incC, 0, sizeof incC);
    for (int h = 0; h < n2; ++h) ks64(incB + 64 * h);
    for (int h = 0; h < n2 && h < 64; ++h) incC[h] = incB[64 * h + 63];
    ks64(incC);
    double run = 0.0;
    for (int g = 0; g < n1; ++g) {
        const int h = g / 64;
        const double eC = h ? incC[h - 1] : 0.0, eB = (g % 64) ? incB[g - 1] : 0.0;
        U->E[g] = eC + eB;
        const double W = U->E[g] + TT[g];
        if (W > run) run = W;
        U->CM[g] = run;
    }
    U->S = run;
    U->valid = (run > 0.0) && (run < INFINITY);
    free(KG); free(TT); free(incB);
}
static void upper_free(upper_t* U) { free(U->e); free(U->sc); free(U->m); free(U->E); free(U->sig); free(U->CM); }

/* #{k : num_k < tau}: the searchsorted(side='left') of src/Filtering.py:34 / src/PGAS.py:122 */
static int64_t cdf_count(const upper_t* U, const uint64_t* c, int64_t N, double tau) {
    int g = 0;
    while (g < U->n1 && U->CM[g] < tau) ++g;
    if (g >= U->n1) return N;
    const double cp = g ? U->CM[g - 1] : 0.0, E = U->E[g], sg = U->sig[g];
    int b = g * PGAS_GRP;
    for (;; ++b) { /* the last segment of the group reaches W_g >= tau */
        double v = E + sg * U->m[b];
        if (v < cp) v = cp;
        if (!(v < tau)) break;
    }
    const double mp = (b % PGAS_GRP) ? U->m[b - 1] : 0.0;
    int64_t base = (int64_t)b * PGAS_SEG;
    int64_t n = N - base < PGAS_SEG ? N - base : PGAS_SEG;
    int64_t cnt = 0;
    for (int64_t k = 0; k < n; ++k) {
        double v = U->e[b] + U->sc[b] * (pgas_u64_to_double(c[base + k]) * PGAS_FIX_INV);
        if (v < mp) v = mp;
        double num = E + sg * v;
        if (num < cp) num = cp;
        if (num < tau) ++cnt; /* monotone in k, so this is a count of a prefix */
    }
    return base + cnt;
}

/* ---- pieces of the resampling machinery, exported for the sharded-decomposition test (tests/test_sharded_cpu.py):
 * per-segment partials, and the search of a slot range given ALL segments' partials (as they are after an all-gather). */
EXPORT void oc_segment_partials(const double* lw, int64_t n, double* segm, uint64_t* segs, uint64_t* c) {
    int nseg = (int)((n + PGAS_SEG - 1) / PGAS_SEG);
    for (int b = 0; b < nseg; ++b) {
        int64_t base = (int64_t)b * PGAS_SEG;
        int len = n - base < PGAS_SEG ? (int)(n - base) : PGAS_SEG;
        segment_scan(lw + base, len, &segm[b], c + base, &segs[b]);
    }
}
EXPORT void oc_resample_range(int32_t nseg, const double* segm, const uint64_t* segs, const uint64_t* c, int64_t N, double u,
                              int64_t i0, int64_t i1, int32_t* anc) {
    upper_t U;
    upper_build(&U, nseg, segm, segs);
    for (int64_t i = i0; i < i1; ++i) {
        int64_t a = i;
        if (U.valid) {
            double Ui = (u + (double)i) / (double)N;
            a = cdf_count(&U, c, N, Ui * U.S);
            if (a > N - 1) a = N - 1;
        }
        anc[i - i0] = (int32_t)a;
    }
    upper_free(&U);
}

/* ------------------------------------------------------------------------------- step -- */
/*
 * One conditional-SMC step, reference src/PGAS.py:79-153 (quirks Q1, Q3, Q4, Q6 reproduced).
 *   x_prev (N,nx), logw_prev (N) or NULL (= zeros), A (nx,M), LS = chol(S) (nx,nx lower),
 *   LSinv = LS^-1, cS = -nx/2 log(2pi) - sum log diag LS, ref_t (nx)
 * outputs: logw_new (N), x_new (N,nx), anc (N int32); optional debug arrays may be NULL.
 */
EXPORT int oc_step(const oc_model* m, int32_t t, uint64_t seed, const double* x_prev,
                   const double* logw_prev, const double* A, const double* LS, const double* LSinv,
                   double cS, const double* ref_t, double* logw_new, double* x_new, int32_t* anc,
                   double* dbg_laux, double* dbg_lw1, double* dbg_lw2, double* dbg_aux,
                   double* dbg_u /* [u_resample, u_ancestor, S1, S2] */) {
    const int N = m->N, nx = m->nx;
    const int nseg = (N + PGAS_SEG - 1) / PGAS_SEG;
    int64_t g = grid_size(m);
    double* G = (double*)malloc(sizeof(double) * (size_t)(nx * g));
    oc_pack_coeff(m, A, G);
    double* laux = (double*)malloc(sizeof(double) * N);
    double* auxall = m->corrected ? (double*)malloc(sizeof(double) * (size_t)N * nx) : NULL;
    double* lw1 = (double*)malloc(sizeof(double) * N);
    double* lw2 = (double*)malloc(sizeof(double) * N);
    uint64_t* c1 = (uint64_t*)malloc(sizeof(uint64_t) * N);
    uint64_t* c2 = (uint64_t*)malloc(sizeof(uint64_t) * N);
    double* segm1 = (double*)malloc(sizeof(double) * nseg);
    double* segm2 = (double*)malloc(sizeof(double) * nseg);
    uint64_t* segs1 = (uint64_t*)malloc(sizeof(uint64_t) * nseg);
    uint64_t* segs2 = (uint64_t*)malloc(sizeof(uint64_t) * nseg);
    const double* yt = m->y + (size_t)t * m->ny;
    const double* ut = m->u + (size_t)t * m->nu;

    for (int p = 0; p < N; ++p) {
        double aux[PGAS_MAX_NX], z[PGAS_MAX_NX + 1], dq[PGAS_MAX_NX], quad = 0.0;
        eval_mean(m, G, x_prev + (size_t)p * nx, ut, aux);              /* :52-55 */
        laux[p] = loglik(m, yt, aux);                                   /* :93-100 */
        lw1[p] = laux[p] + (logw_prev ? logw_prev[p] : 0.0);            /* :101 */
        for (int k = 0; k < nx; ++k) dq[k] = ref_t[k] - aux[k];         /* :109-116 */
        for (int k = 0; k < nx; ++k) {
            double w = 0.0;
            for (int l = 0; l <= k; ++l) w = PGAS_FMA(LSinv[k * nx + l], dq[l], w);
            quad = PGAS_FMA(w, w, quad);
        }
        lw2[p] = lw1[p] + PGAS_FMA(-0.5, quad, cS);                     /* :117 */
        if (m->corrected) {
            memcpy(auxall + (size_t)p * nx, aux, sizeof(double) * nx);  /* the draw waits for the ancestors */
        } else {
            pgas_rng_normals(seed, PGAS_STREAM_PROP, (uint32_t)t, (uint64_t)p, nx, z);
            for (int k = 0; k < nx; ++k) {                              /* :130-133, Q1: own state, not state[a] */
                double xv = aux[k];
                for (int l = 0; l <= k; ++l) xv = PGAS_FMA(LS[k * nx + l], z[l], xv);
                x_new[(size_t)p * nx + k] = xv;
            }
        }
        if (dbg_aux) memcpy(dbg_aux + (size_t)p * nx, aux, sizeof(double) * nx);
    }
    if (!m->corrected) memcpy(x_new + (size_t)(N - 1) * nx, ref_t, sizeof(double) * nx);   /* :134 */

    for (int b = 0; b < nseg; ++b) {
        int base = b * PGAS_SEG, n = N - base < PGAS_SEG ? N - base : PGAS_SEG;
        segment_scan(lw1 + base, n, &segm1[b], c1 + base, &segs1[b]);
        segment_scan(lw2 + base, n, &segm2[b], c2 + base, &segs2[b]);
    }
    upper_t U1, U2;
    upper_build(&U1, nseg, segm1, segs1);
    upper_build(&U2, nseg, segm2, segs2);

    double u1 = pgas_rng_uniform(seed, PGAS_STREAM_RESAMPLE, (uint32_t)t);
    double u2 = pgas_rng_uniform(seed, PGAS_STREAM_ANCESTOR, (uint32_t)t);
    for (int i = 0; i < N; ++i) {                                       /* Filtering.py:28-35 */
        int64_t a = i;
        if (U1.valid) {
            double Ui = (u1 + (double)i) / (double)N;
            a = cdf_count(&U1, c1, N, Ui * U1.S);
            if (a > N - 1) a = N - 1;
        }
        anc[i] = (int32_t)a;
    }
    {                                                                   /* :121-127, Q4 */
        int64_t r = N - 1;
        if (U2.valid) {
            r = cdf_count(&U2, c2, N, u2 * U2.S);
            if (r > N - 1) r = N - 1;
        }
        anc[N - 1] = (int32_t)r;
    }
    if (m->corrected) {   /* corrected mode: x_new_i = aux[a_i] + L_S z_i, same noise as the default mode */
        for (int p = 0; p < N; ++p) {
            double z[PGAS_MAX_NX + 1];
            pgas_rng_normals(seed, PGAS_STREAM_PROP, (uint32_t)t, (uint64_t)p, nx, z);
            for (int k = 0; k < nx; ++k) {
                double xv = auxall[(size_t)anc[p] * nx + k];
                for (int l = 0; l <= k; ++l) xv = PGAS_FMA(LS[k * nx + l], z[l], xv);
                x_new[(size_t)p * nx + k] = xv;
            }
        }
        memcpy(x_new + (size_t)(N - 1) * nx, ref_t, sizeof(double) * nx);
    }
    for (int p = 0; p < N; ++p)                                         /* :137-147 */
        logw_new[p] = loglik(m, yt, x_new + (size_t)p * nx) - laux[anc[p]];

    if (dbg_laux) memcpy(dbg_laux, laux, sizeof(double) * N);
    if (dbg_lw1) memcpy(dbg_lw1, lw1, sizeof(double) * N);
    if (dbg_lw2) memcpy(dbg_lw2, lw2, sizeof(double) * N);
    if (dbg_u) { dbg_u[0] = u1; dbg_u[1] = u2; dbg_u[2] = U1.S; dbg_u[3] = U2.S; }
    upper_free(&U1); upper_free(&U2);
    free(G); free(laux); free(lw1); free(lw2); free(c1); free(c2);
    free(segm1); free(segm2); free(segs1); free(segs2); free(auxall);
    return 0;
}

EXPORT void oc_set_corrected(oc_model* m, int32_t on) { m->corrected = on ? 1 : 0; }

/* x_0 ~ N(m0, P0), conditioned particle last  (src/PGAS.py:155-174, :194) */
EXPORT void oc_init_state(const oc_model* m, uint64_t seed, const double* m0, const double* L0,
                          const double* ref0, double* x0) {
    const int N = m->N, nx = m->nx;
    for (int p = 0; p < N; ++p) {
        double z[PGAS_MAX_NX + 1];
        pgas_rng_normals(seed, PGAS_STREAM_INIT, 0u, (uint64_t)p, nx, z);
        for (int k = 0; k < nx; ++k) {
            double xv = m0[k];
            for (int l = 0; l <= k; ++l) xv = PGAS_FMA(L0[k * nx + l], z[l], xv);
            x0[(size_t)p * nx + k] = xv;
        }
    }
    memcpy(x0 + (size_t)(N - 1) * nx, ref0, sizeof(double) * nx);
}

/* idx ~ Cat(softmax(logw))  (src/PGAS.py:224-225, Q7) */
EXPORT int64_t oc_final_index(const oc_model* m, uint64_t seed, const double* logw) {
    const int N = m->N;
    const int nseg = (N + PGAS_SEG - 1) / PGAS_SEG;
    uint64_t* c = (uint64_t*)malloc(sizeof(uint64_t) * N);
    double* segm = (double*)malloc(sizeof(double) * nseg);
    uint64_t* segs = (uint64_t*)malloc(sizeof(uint64_t) * nseg);
    for (int b = 0; b < nseg; ++b) {
        int base = b * PGAS_SEG, n = N - base < PGAS_SEG ? N - base : PGAS_SEG;
        segment_scan(logw + base, n, &segm[b], c + base, &segs[b]);
    }
    upper_t U;
    upper_build(&U, nseg, segm, segs);
    int64_t r = N - 1;
    if (U.valid) {
        r = cdf_count(&U, c, N, pgas_rng_uniform(seed, PGAS_STREAM_FINAL, 0u) * U.S);
        if (r > N - 1) r = N - 1;
    }
    upper_free(&U);
    free(c); free(segm); free(segs);
    return r;
}

/*
 * Whole sweep, reference src/PGAS.py:176-228.  traj (T,nx) out.  If x_trace (T,N,nx),
 * anc_trace (T-1,N int32), logw_last (N) are non-NULL they receive the traces; otherwise only
 * two time slices are kept (used by the CPU-baseline timing at large N... the back-trace then
 * needs the traces, so they are required unless traj is NULL).
 * nsteps_limit > 0 stops after that many steps (timing sample); traj must then be NULL.
 */
EXPORT int oc_sweep(const oc_model* m, uint64_t seed, const double* ref, const double* A,
                    const double* LS, const double* LSinv, double cS, const double* m0,
                    const double* L0, double* traj, double* x_trace, int32_t* anc_trace,
                    double* logw_last, int32_t nsteps_limit) {
    const int N = m->N, T = m->T, nx = m->nx;
    const size_t row = (size_t)N * nx;
    int keep = (x_trace != NULL && anc_trace != NULL);
    if (traj && !keep) return 1;
    double* xa = keep ? NULL : (double*)malloc(sizeof(double) * row);
    double* xb = keep ? NULL : (double*)malloc(sizeof(double) * row);
    int32_t* ab = keep ? NULL : (int32_t*)malloc(sizeof(int32_t) * N);
    double* lwa = (double*)malloc(sizeof(double) * N);
    double* lwb = (double*)malloc(sizeof(double) * N);
    double* xp = keep ? x_trace : xa;
    oc_init_state(m, seed, m0, L0, ref, xp);
    const double* lwp = NULL;
    int tlast = (nsteps_limit > 0 && nsteps_limit < T - 1) ? nsteps_limit : T - 1;
    for (int t = 1; t <= tlast; ++t) {
        double* xn = keep ? x_trace + (size_t)t * row : ((t & 1) ? xb : xa);
        int32_t* an = keep ? anc_trace + (size_t)(t - 1) * N : ab;
        double* lwn = (t & 1) ? lwb : lwa;
        oc_step(m, t, seed, xp, lwp, A, LS, LSinv, cS, ref + (size_t)t * nx, lwn, xn, an, NULL, NULL,
                NULL, NULL, NULL);
        xp = xn;
        lwp = lwn;
    }
    if (logw_last && lwp) memcpy(logw_last, lwp, sizeof(double) * N);
    if (traj) {
        int64_t b = oc_final_index(m, seed, lwp);
        memcpy(traj + (size_t)(T - 1) * nx, x_trace + (size_t)(T - 1) * row + (size_t)b * nx,
               sizeof(double) * nx);
        for (int i = T - 2; i >= 0; --i) { /* src/Filtering.py:51-53 */
            b = anc_trace[(size_t)i * N + b];
            memcpy(traj + (size_t)i * nx, x_trace + (size_t)i * row + (size_t)b * nx, sizeof(double) * nx);
        }
    }
    free(xa); free(xb); free(ab); free(lwa); free(lwb);
    return 0;
}
