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

/* Three small fusions of the filter step's per-particle glue (each replaces a handful of elementwise launches):
 *   pgas_m_rng_student_t_df  Student-t variates with nu[p] = nu0 + nu_scale * src[anc[p]] formed in the kernel (df = P3 + lambda T3[a], BI:45);
 *   pgas_m_mniw_draw         xi[p] = m[p] + sqrt((P2 + scale T2[a] - q[a]) / (P3 + scale T3[a])) t[p] sqrt(c[p] + 1), a = anc[p]  (BI:64-108, n = 1);
 *   pgas_m_hilbert_basis     phi (n, M) = prod_d sqrt(1/L_d) sin(pi j[m][d] (v_d / div_d - center_d + L_d) / size_d), v = concat(state[p], input)[sel]
 *                            (src/BasisFunctions.py:77-80 under the vmap of src/Algorithm1.py:220-225): sel / div / center / L / size host (D <= 4),
 *                            idx_dev (M, D) int32 on the device. */
int pgas_m_rng_student_t_df(pgas_ctx* ctx, uint64_t seed, uint32_t stream, uint32_t t, int64_t p0, int64_t n, const int32_t* anc_dev, const double* src_dev,
                            double nu0, double nu_scale, double* out_dev, void* stream_handle);
int pgas_m_mniw_draw(pgas_ctx* ctx, int64_t n, double scale, const int32_t* anc_dev, const double* m_dev, const double* c_dev, const double* q_dev,
                     const double* T2_dev, const double* T3_dev, double P2, double P3, const double* t_dev, double* out_dev, void* stream_handle);
/* g[p] = lbm(P + T[p]) - lbm(P + T[p] + R) with lbm(nu, Psi, logdet) = -M/2 log 2pi + logdet/2 - nu/2 log 2 - lgamma(nu/2) + nu/2 log Psi: the
 * base-measure term of the conditional filter's ancestor weights (src/Algorithm3.py:93-108, BI:111-124, n = 1) from the q / logdet of the two
 * pgas_m_mniw_solve calls; r2_dev / r3_dev: the reference statistics' T2 / T3 (one double each, on the device). */
int pgas_m_lbm_diff(pgas_ctx* ctx, int64_t n, int32_t M, const double* T2_dev, const double* T3_dev, const double* q1_dev, const double* logdet1_dev,
                    const double* q2_dev, const double* logdet2_dev, double P2, double P3, const double* r2_dev, const double* r3_dev, double* out_dev,
                    void* stream_handle);
int pgas_m_hilbert_basis(pgas_ctx* ctx, int64_t n, int32_t M, int32_t D, const double* state_dev, int32_t nx, const double* input_dev, int32_t nu,
                         const int32_t* sel, const double* div, const double* center, const double* L, const double* size, const int32_t* idx_dev,
                         double* out_dev, void* stream_handle);

/* A model callable as data (src/StateSpaceModel.py:32-87: transition_mdl / output_mdl / draw_state / log_likelihood, which the reference
 * evaluates per particle under jax.vmap): `code_dev` (ninstr, 4) int32 = (opcode, destination, source a, source b) over `nreg` <= 96 registers
 * per particle; registers [0, n_in) are preloaded with the state's nx columns, the nu input components and the interface variables'
 * components (in that order), [n_in, n_in + nconst) with `consts_dev`; opcodes 1..14 = add sub mul div neg cos sin tan tanh atan sqrt exp
 * sign mov (pgas_amd/exprs.py traces a model written against an array namespace into this form).  The registers `out_regs` (host, nout <= 8)
 * hold the result v; mode 0: out (n, nout) = v; mode 1: out = v + aux (n, nout) mat^T (draw_state: aux standard normals, mat = chol Q);
 * mode 2: out (n) = cR - |mat (aux (nout) - v)|^2 / 2 (log_likelihood: aux = y_t, mat = chol(R)^-1).  anc_dev (nullable): particle p reads
 * the state and interface variables of particle anc[p].  iv_dev / iv_widths: host arrays of n_iv <= 4 device pointers / widths. */
int pgas_m_expr_eval(pgas_ctx* ctx, int64_t n, const int32_t* code_dev, int32_t ninstr, const double* consts_dev, int32_t nconst, int32_t n_in,
                     int32_t nreg, const int32_t* out_regs, int32_t nout, const double* state_dev, int32_t nx, const int32_t* anc_dev,
                     const double* input_dev, int32_t nu, const double* const* iv_dev, const int32_t* iv_widths, int32_t n_iv, int32_t mode,
                     const double* aux_dev, const double* mat_dev, double cR, double* out_dev, void* stream_handle);

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
