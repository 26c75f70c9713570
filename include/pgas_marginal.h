/* pgas_marginal.h -- C ABI of the device primitives behind the MARGINALISED family (reference src/Algorithm1.py,
 * src/Algorithm3.py, src/Algorithm2.py).  Same conventions as pgas_hip.h: return 0 = ok, error
 * text through pgas_last_error(ctx), device pointers of fp64 / int32 arrays, work enqueued on the caller's stream.  `ctx` is
 * any context of the target device (the host mirror uses the utility context of pgas_amd/_lib.py); it supplies the device and
 * the error channel only.
 *
 * The reference maps per-particle functions with jax.vmap; these entry points are the batched bodies of
 *   jax.random.normal / jax.random.t                                    src/StateSpaceModel.py:67, src/BayesianInferrence.py:104
 *   BI.prior_mniw_mean + einsum            (auxiliary interface variable) src/Algorithm1.py:211-231
 *   BI.prior_mniw_2naturalPara_inv + BI.prior_mniw_Predictive            src/Algorithm1.py:251-262, BI:35-45, :64-89
 *   BI.prior_mniw_log_base_measure                                       src/Algorithm3.py:95-108, BI:111-124
 *   forgetting, ancestor gather and BI.prior_mniw_calcStatistics update  src/Algorithm1.py:317-320, :358-377
 * for a scalar interface variable (n = 1: every instantiation in the reference). */
#ifndef PGAS_MARGINAL_H
#define PGAS_MARGINAL_H

#include "pgas_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* The scalar uniform of (seed, stream, t): host value, identical to what the device kernels would draw (pgas_canon.h). */
double pgas_m_rng_uniform(uint64_t seed, uint32_t stream, uint32_t t);

/* out (n, ncol): normals of particles p0 .. p0+n at (stream, t); ncol <= 8. */
int pgas_m_rng_normal(pgas_ctx* ctx, uint64_t seed, uint32_t stream, uint32_t t, int64_t p0, int64_t n, int32_t ncol,
                      double* out_dev, void* stream_handle);

/* out (n): Student-t(nu[p]) variates (z / sqrt(chi2_nu / nu), Marsaglia-Tsang gamma sampler on the same Philox streams). */
/* Device-resident time index.  With t_dev != NULL every later pgas_m_rng_normal / _student_t / _uniform_dev call of this context takes
 * the time index of its Philox counters from *t_dev at execution time (the `t` argument is ignored): a filter step captured once in a HIP
 * graph can then be replayed for every t (reference loop src/Algorithm1.py:418-457).  NULL restores the argument.  t_dev must stay valid. */
int pgas_m_set_time_source(pgas_ctx* ctx, const uint32_t* t_dev);
/* out_dev[0] = the uniform pgas_m_rng_uniform(seed, stream, t) returns on the host, written on the device (t from the time source if set) */
int pgas_m_rng_uniform_dev(pgas_ctx* ctx, uint64_t seed, uint32_t stream, uint32_t t, double* out_dev, void* stream_handle);

int pgas_m_rng_student_t(pgas_ctx* ctx, uint64_t seed, uint32_t stream, uint32_t t, int64_t p0, int64_t n, const double* nu_dev,
                         double* out_dev, void* stream_handle);
/* The same Student-t variates computed on the HOST by the library's own arithmetic (no context, no device; bit-identical to the
 * kernel's): for host-side helpers such as prior_mniw_drawPred (BI:92-108), so that one key never means two samplers. */
int pgas_m_rng_student_t_host(uint64_t seed, uint32_t stream, uint32_t t, int64_t p0, int64_t n, const double* nu_host, double* out_host);

/* chi^2(nu_p) variates, out[p] = 2 Gamma(nu_p / 2) on the counters of particle p0 + p (pgas_rng_gamma, include/pgas_canon.h): the
 * diagonal of the Bartlett factor in PGAS.sample_params (reference src/PGAS.py:323-327, jax.random.chisquare there). */
int pgas_m_rng_chi2(pgas_ctx* ctx, uint64_t seed, uint32_t stream, uint32_t t, int64_t p0, int64_t n, const double* nu_dev, double* out_dev,
                    void* stream_handle);

/* Per particle p, with s = anc[p] (anc NULL: s = p): eta0 = P0 + scale T0[s] (+ R0), eta1 = P1 + scale T1[s] (+ R1)
 * (M <= 126, eta1 symmetric positive definite; M <= 62: one matrix row per lane, the trailing update on the f64 matrix cores; 63 ... 126: two rows
 * per lane, the triangle updated in place in LDS -- a generality path);
 *   m[p] = eta0^T eta1^-1 phi[p],  c[p] = phi[p]^T eta1^-1 phi[p],  q[p] = eta0^T eta1^-1 eta0,  logdet[p] = log det eta1.
 * R0/R1 (the reference trajectory's statistics, src/Algorithm3.py:96-101), phi and every output may be NULL.
 * Lfac_dev (n, (M+2)(M+3)/2), optional, receives the packed Cholesky factor for pgas_m_mniw_trisolve: rows of L with 1/L_kk on the
 * diagonal, followed by the two right-hand-side rows the kernel eliminates along with the matrix (row M = L^-1 phi, row M+1 =
 * w = L^-1 eta0).  The children of a resampled particle share its matrix (src/Algorithm1.py:358-361), so it is factorised once
 * per step, before resampling, not once more per child.
 * Asynchronous; a matrix that is not positive definite is counted on the device and reported by pgas_m_check. */
int pgas_m_mniw_solve(pgas_ctx* ctx, int64_t n, int32_t M, double scale, const int32_t* anc_dev, const double* P0_dev, const double* P1_dev,
                      const double* T0_dev, const double* T1_dev, const double* R0_dev, const double* R1_dev, const double* phi_dev,
                      double* m_dev, double* c_dev, double* q_dev, double* logdet_dev, double* Lfac_dev, void* stream_handle);

/* With the factor stored by pgas_m_mniw_solve: m[p] = w[s] . v, c[p] = v . v, v = L[s]^-1 phi[p], s = anc[p] (NULL: p). */
int pgas_m_mniw_trisolve(pgas_ctx* ctx, int64_t n, int32_t M, const int32_t* anc_dev, const double* Lfac_dev, const double* phi_dev,
                         double* m_dev, double* c_dev, void* stream_handle);

/* Synchronises the stream; PGAS_E_STATE if any pgas_m_mniw_solve since the last check met a matrix that was not positive
 * definite (its outputs are NaN). */
int pgas_m_check(pgas_ctx* ctx, void* stream_handle);

/* T_out[p] = scale * T_in[anc[p]] + (phi[p] xi[p], phi[p] phi[p]^T, xi[p]^2, 1); anc may be NULL (identity).  In and out must
 * not alias. */
int pgas_m_stats_gather_update(pgas_ctx* ctx, int64_t n, int32_t M, double scale, const int32_t* anc_dev, const double* T0_in,
                               const double* T1_in, const double* T2_in, const double* T3_in, const double* phi_dev,
                               const double* xi_dev, double* T0_out, double* T1_out, double* T2_out, double* T3_out,
                               void* stream_handle);

/* S = sum_p w[p] (T0[p], T1[p], T2[p], T3[p]): the weighted statistics trace of src/Algorithm1.py:166-170, :445-457
 * (S0 (M), S1 (M,M), S2 (1), S3 (1)).  Streaming reduction in two deterministic passes. */
int pgas_m_weighted_stats(pgas_ctx* ctx, int64_t n, int32_t M, const double* w_dev, const double* T0_dev, const double* T1_dev,
                          const double* T2_dev, const double* T3_dev, double* S0_dev, double* S1_dev, double* S2_dev, double* S3_dev,
                          void* stream_handle);

/* The same four operations for an interface variable with nvar > 1 components (the reference's formulas are general in n: eta0 (M, n),
 * eta2 (n, n), BI:18-50, 53-61, 64-108; no configuration of the reference uses n > 1).  Layouts: P0 / R0 (M, nvar), T0 (n, M, nvar),
 * T2 (n, nvar, nvar), xi (n, nvar), all row-major; results m (n, nvar) = eta0^T eta1^-1 phi (BI:81), q (n, nvar, nvar) = eta0^T eta1^-1 eta0
 * (row_scale = eta2 - q, BI:42), c and logdet as above; Lfac (n, (M+1+nvar)(M+2+nvar)/2) carries nvar right-hand-side rows behind the
 * factor.  M + 1 + nvar <= 128, nvar <= 8.  nvar = 1 is exactly the scalar entry point (which forwards here). */
int pgas_m_mniw_solve_n(pgas_ctx* ctx, int64_t n, int32_t M, int32_t nvar, double scale, const int32_t* anc_dev, const double* P0_dev,
                        const double* P1_dev, const double* T0_dev, const double* T1_dev, const double* R0_dev, const double* R1_dev,
                        const double* phi_dev, double* m_dev, double* c_dev, double* q_dev, double* logdet_dev, double* Lfac_dev,
                        void* stream_handle);
int pgas_m_mniw_trisolve_n(pgas_ctx* ctx, int64_t n, int32_t M, int32_t nvar, const int32_t* anc_dev, const double* Lfac_dev,
                           const double* phi_dev, double* m_dev, double* c_dev, void* stream_handle);
int pgas_m_stats_gather_update_n(pgas_ctx* ctx, int64_t n, int32_t M, int32_t nvar, double scale, const int32_t* anc_dev, const double* T0_in,
                                 const double* T1_in, const double* T2_in, const double* T3_in, const double* phi_dev,
                                 const double* xi_dev, double* T0_out, double* T1_out, double* T2_out, double* T3_out,
                                 void* stream_handle);
int pgas_m_weighted_stats_n(pgas_ctx* ctx, int64_t n, int32_t M, int32_t nvar, const double* w_dev, const double* T0_dev,
                            const double* T1_dev, const double* T2_dev, const double* T3_dev, double* S0_dev, double* S1_dev,
                            double* S2_dev, double* S3_dev, void* stream_handle);

#ifdef __cplusplus
}
#endif
#endif /* PGAS_MARGINAL_H */
