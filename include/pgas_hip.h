/*
 * pgas_hip.h -- C ABI of libpgas_hip.so, the MI355X (gfx950) engine for the conditional-SMC /
 * PGAS hot path of VolkmannB/bayesian-inference-with-explicit-and-implicit-prior-knowledge.
 *
 * The reference has no FFI: its boundary is the Python call surface of src/PGAS.py.  Each entry
 * point below names the reference interface it stands under (paths relative to the reference
 * root); INTEGRATION.md shows the ctypes binding a maintainer would add.
 *
 * Conventions
 *   - plain C, no C++ types, no torch types; every function returns 0 on success or a negative
 *     PGAS_E_* code, and pgas_last_error(ctx) gives the message.  No exceptions cross the ABI.
 *   - "dev" pointers are device (HBM) pointers on ctx's device, "host" pointers are ordinary
 *     host memory read during the call.  The caller owns every buffer it passes; the library
 *     owns only the context (model tables, traces, scan scratch).
 *   - all work is enqueued on the caller's stream (`stream` = hipStream_t, e.g.
 *     torch.cuda.current_stream().cuda_stream); calls return without synchronising unless noted.
 *   - one context per device; a context is not thread-safe, distinct contexts are independent.
 *   - arithmetic is fp64 throughout (reference src/__init__.py:4) in the canonical order of
 *     DESIGN.md section 4; ancestor indices are int32.
 *   - particle arrays are row-major (N, nx) exactly as the reference's state_trace[t]
 *     (src/PGAS.py:160-162).
 */
#ifndef PGAS_HIP_H
#define PGAS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PGAS_OK 0
#define PGAS_E_ARG (-1)      /* bad argument / unsupported model shape */
#define PGAS_E_HIP (-2)      /* a HIP runtime call failed */
#define PGAS_E_STATE (-3)    /* call sequence error (e.g. sweep before set_params) */
#define PGAS_E_NOMEM (-4)    /* device allocation failed */

typedef struct pgas_ctx pgas_ctx;

/*
 * Declarative model description.  The reference passes Python callables that JAX traces
 * (basis_fcn, likelihood_fcn: src/PGAS.py:20-21,31-32); a HIP kernel cannot, so the two families
 * the reference actually instantiates are described by tables:
 *
 *   basis_fcn(state, input)  = Hilbert-space GP basis of src/BasisFunctions.py:8-80 evaluated at
 *        v[sel[d]] with v = concat(state, input):   phi_m = nrm * prod_d sin(pi * idx[m][d] * r_d),
 *        r_d = v[sel[d]] * alpha[d] + beta[d]
 *        (alpha_d = 1/(div_d * 2 L_d), beta_d = (L_d - center_d)/(2 L_d), nrm = prod_d L_d^-1/2;
 *         covers src/Toy_Example.py:146, src/EMPS.py:110-113, src/SingleMassOscillator.py:151).
 *   likelihood_fcn(obs, state, input) = log N(obs; H state, R)
 *        (src/Toy_Example.py:142-144, src/EMPS.py:250-252), given as H, LRinv = chol(R)^-1 and
 *        cR = -ny/2 log(2 pi) - sum(log diag chol(R)).
 */
typedef struct pgas_model_desc {
    int32_t N;            /* particles            (N_samples,   src/PGAS.py:26)  */
    int32_t T;            /* time steps           (observations.shape[0], :159)  */
    int32_t nx, ny, nu;   /* state / observation / input dimensions (nx<=2, ny<=2, nu<=4) */
    int32_t M, D;         /* basis functions, basis input dimension (D<=3) */
    const int32_t* idx;   /* host (M,D): frequency j of basis m in dimension d, reference order */
    const int32_t* sel;   /* host (D)  */
    const double* alpha;  /* host (D)  */
    const double* beta;   /* host (D)  */
    double nrm;
    const double* H;      /* host (ny,nx) */
    const double* LRinv;  /* host (ny,ny) lower triangular */
    double cR;
    const double* m0;     /* host (nx)      init_state_mean (src/PGAS.py:29) */
    const double* L0;     /* host (nx,nx)   chol(init_state_cov) lower (src/PGAS.py:30) */
    const double* y;      /* host (T,ny)    observations */
    const double* u;      /* host (T,nu)    inputs; may be NULL when nu == 0 */
    int32_t device;       /* HIP device ordinal */
    int32_t keep_logw_trace; /* 1: keep the (T,N) log-weight trace like src/PGAS.py:163 (costs 8 B/particle-step) */
    int32_t no_fast_variant; /* test knob: 1 = never use the k_propagate instantiations specialised for the reference's model shapes */
} pgas_model_desc;

/* Replaces condSequentialMonteCarlo.__init__ (src/PGAS.py:24-43) / PGAS.__init__ (:237-260):
 * copies the model tables to the device.  Traces are allocated lazily by the first sweep. */
int pgas_create(const pgas_model_desc* desc, pgas_ctx** out);
void pgas_destroy(pgas_ctx* ctx);
/* message of the last failing call on ctx (or of the last failing pgas_create when ctx == NULL) */
const char* pgas_last_error(const pgas_ctx* ctx);

/* Canonical-arithmetic constants the build was made with (segment length, fixed-point bits). */
int32_t pgas_segment_size(void);

/* Parameters of the transition x_t ~ N(A phi(x_{t-1},u_t), S) for the following step/sweep calls
 * (coeff_mat, error_cov of src/PGAS.py:85-86,180-181).  A_dev: device (nx,M) row-major.
 * LS_host = chol(S) lower (nx,nx), LSinv_host = LS^-1, cS = -nx/2 log(2 pi) - sum(log diag LS). */
int pgas_set_params(pgas_ctx* ctx, const double* A_dev, const double* LS_host, const double* LSinv_host,
                    double cS, void* stream);

/* The same with error_cov on the DEVICE: S_dev (nx,nx) row-major, symmetric positive definite.  Its Cholesky factor, the factor's
 * inverse and the constant are formed by the pack kernel (IEEE sqrt, /, *, - and the shared logarithm, in a fixed order), so a Gibbs
 * iteration -- PGAS.sample_params -> sweep, src/PGAS.py:366-378 -- needs no host round trip.  pgas_get_params reads back the factor,
 * its inverse and the constant the kernels use (synchronises the stream; for tests and for replaying a chain with the oracle). */
int pgas_set_params_dev(pgas_ctx* ctx, const double* A_dev, const double* S_dev, void* stream);
int pgas_get_params(pgas_ctx* ctx, double* LS_host, double* LSinv_host, double* cS, void* stream);

/* Test hook: phi (np,M) = basis_fcn(x[p], inputs[t]) in reference order
 * (vmap(basis_fcn) of src/PGAS.py:52-54). x_dev (np,nx). */
int pgas_basis_eval(pgas_ctx* ctx, const double* x_dev, int64_t np, int32_t t, double* phi_dev, void* stream);

/* Test hook: aux (N,nx) = _generate_auxiliary_states (src/PGAS.py:45-57). */
int pgas_aux_states(pgas_ctx* ctx, const double* x_dev, int32_t t, double* aux_dev, void* stream);

/* condSequentialMonteCarlo._init_algorithm + the conditioning of :194: x0 (N,nx) ~ N(m0,P0),
 * x0[N-1] = ref0 (host, nx). */
int pgas_init_state(pgas_ctx* ctx, uint64_t seed, const double* ref0_host, double* x0_dev, void* stream);

/* condSequentialMonteCarlo.step (src/PGAS.py:79-153): one conditional-SMC step at `time` = t.
 * logw_dev (N) may be NULL (= zeros, the t = 1 case of :163); ref_t_host (nx).
 * Outputs: logw_new_dev (N), x_new_dev (N,nx), anc_dev (N int32 = a_indices). */
int pgas_step(pgas_ctx* ctx, int32_t t, uint64_t seed, const double* logw_dev, const double* x_dev,
              const double* ref_t_host, double* logw_new_dev, double* x_new_dev, int32_t* anc_dev, void* stream);

/* condSequentialMonteCarlo.__call__ (src/PGAS.py:176-228): the whole sweep incl. the final index
 * draw (:224-225) and the back-trace (src/Filtering.py:40-55), all on the device.
 * ref_dev (T,nx) in, traj_dev (T,nx) out. */
int pgas_sweep(pgas_ctx* ctx, uint64_t seed, const double* ref_dev, double* traj_dev, void* stream);

/* Device pointers of the traces of the last sweep: x_trace (T,N,nx), anc_trace (T-1,N) int32,
 * logw_last (N) = log_weights_trace[T-1], logw_trace (T,N) or NULL.  Valid until the next sweep.
 * x_trace / anc_trace are single arrays on an unsharded context (state_trace / ancestor_trace of src/PGAS.py:160-164).  A sharded
 * context (and any context after PGAS_OPT_TRACE_BLOCK_BYTES) keeps them as ROW BLOCKS -- consecutive time rows in allocations of at
 * most 1 GiB, so that each can cross process boundaries as one HIP IPC handle -- and refuses x_trace / anc_trace here (PGAS_E_STATE):
 * pgas_trace_layout gives the blocking, pgas_trace_row the device pointer of one time row. */
int pgas_get_traces(pgas_ctx* ctx, double** x_trace, int32_t** anc_trace, double** logw_last, double** logw_trace);
#define PGAS_TRACE_X 0    /* T rows of (N, nx) f64: state_trace[t] */
#define PGAS_TRACE_LA 1   /* T rows of nseg*1024 f64: log p(y_t | aux_t)      (hand-off rows between the two pipelines) */
#define PGAS_TRACE_H 2    /* T rows: log N(ref_t; aux_t, S) */
#define PGAS_TRACE_LN 3   /* T rows: log p(y_t | x_t) */
#define PGAS_TRACE_ANC 4  /* T-1 rows of N int32: ancestor_trace[t] */
/* info4 = {rows, rows per block (a power of two; = rows when the trace is one array), blocks, bytes per row} */
int pgas_trace_layout(pgas_ctx* ctx, int32_t kind, int64_t* info4);
int pgas_trace_row(pgas_ctx* ctx, int32_t kind, int32_t t, void** row_dev);

/* Index drawn by the last sweep at src/PGAS.py:225 (synchronises the stream). */
int pgas_last_final_index(pgas_ctx* ctx, int64_t* idx, void* stream);

/* Measurement aid (bench.py): when on, pgas_sweep launches its two per-step kernels -- k_step (resampling
 * search + softmax scans) and k_propagate (all particles through a chunk of time steps) -- with start/stop HIP events
 * attached to the dispatch (hipExtLaunchKernelGGL), which carry the kernel's own begin/end timestamps on the stream it
 * runs on.  pgas_get_profile synchronises and returns, for the last sweep, the counts and summed durations of the launches
 * that were timed: every 16th by default (on = 1), every n-th for on = n > 1, every launch for on < 0 (timing every launch
 * slows the sweep it measures by ~8 %: timed dispatches serialise with their neighbours on the stream).
 * No reference counterpart (the reference has no timing code). */
int pgas_set_profiling(pgas_ctx* ctx, int32_t on);
int pgas_get_profile(pgas_ctx* ctx, int64_t* resample_launches, double* resample_ms, int64_t* propagate_launches,
                     double* propagate_ms, void* stream);
/* What the last sweep launched: info4 = {time steps per k_propagate launch, group-scan placement (0: k_groups launches between the
 * steps, 1: k_step<LOCAL> scans all groups in every workgroup, 2: in k_step's tail by the workgroup that completes a group, 3: no group
 * scans at all -- the one-launch small sweep (k_sweep_duo / k_sweep_small) ran; + 16 when
 * the last sweep replayed the captured HIP graph; + 32 when k_propagate ran its matrix-core form, PGAS_OPT_MFMA_PROPAGATE),
 * padded innermost basis extent JP, particles per basis pass P} -- bench.py labels its kernels from this. */
int pgas_get_launch_info(pgas_ctx* ctx, int32_t* info4);

/* Tuning / test knobs.  PGAS_OPT_PROPAGATE_CHUNK: time steps per k_propagate launch (0 = the whole sweep in one launch). */
#define PGAS_OPT_PROPAGATE_CHUNK 1
#define PGAS_OPT_PROPAGATE_LDS 4 /* bytes of LDS reserved per k_propagate workgroup while overlapping (occupancy cap) */
#define PGAS_OPT_OVERLAP 3 /* 1 (default): weight recursion on an internal stream, concurrent with k_propagate */
#define PGAS_OPT_FORCE_SLOW_RESAMPLE 2 /* 1: never take k_step<LOCAL> (overrides PGAS_OPT_LOCAL_GROUPS; kept for the tests) */
/* 1: CORRECTED mode, not the reference's behaviour.  src/PGAS.py:131-133 propagates every particle from its own previous
 * state (`state`, not `state[a_indices]`) although the ancestors are recorded and used for the weights and the
 * back-trace.  With this option pgas_step / pgas_sweep draw x_t[i] ~ N(A phi(x_{t-1}[a_i]), S) as a particle filter should
 * (and as src/Algorithm1.py:286-292 does).  Same random numbers, same weight formula; the step becomes a serial chain of
 * three launches.  Default 0 = reproduce the reference. */
#define PGAS_OPT_RESAMPLE_BEFORE_PROPAGATE 5
#define PGAS_OPT_LOCAL_GROUPS 7 /* 1: on a single device with <= 1024 segments every k_step workgroup scans all groups itself (k_step<LOCAL>, no k_groups launch between the steps); default 0, the k_groups path is faster */
#define PGAS_OPT_MAX_LEAD 11 /* sweep: k_propagate may run at most this many event groups (PGAS_OPT_EVENT_STRIDE steps each) ahead of the weight recursion; 0 = unbounded */
#define PGAS_OPT_SYRK_SPLITS 10 /* pgas_suffstats: number of row splits of the Z^T Z product (partial slabs summed in split order); 0 = automatic (about two workgroups per CU) */
#define PGAS_OPT_TAIL_GROUPS 9 /* 1: single device only: the group scans ride in k_step's tail (in-launch hand-off to the workgroup that completes a group) instead of k_groups launches; measured slower, default 0 */
#define PGAS_OPT_EVENT_STRIDE 8 /* k_propagate launches per event that gates the weight recursion's stream (default 8) */
#define PGAS_OPT_SMALL_SWEEP 14 /* 1 (default): a context of at most one segment of particles (N <= 1024 -- the reference's own operating point is N = 200) runs the whole sweep, x_0 to back-trace, as ONE launch instead of ~3 dependent launches per time step: two workgroups (k_sweep_duo), one propagating every step ahead into a ring in device memory, the other running the weight recursion behind it with both fixed-point CDFs, the auxiliary log-likelihoods and the scan scratch in LDS; the propagation noise comes from one grid-wide launch in front of it.  2: the same work on one workgroup, all waves in lock step (k_sweep_small); 0: the general multi-launch path at every size.  Bit-identical results */
#define PGAS_OPT_MFMA_PROPAGATE 15 /* 1: models with a 3-D basis, n_x = 2 and the 729-function index ball of the 11 x 11 x 11 grid (EMPS / Vehicle, src/EMPS.py:101-113) run the innermost sum of k_propagate's contraction A phi(x) (src/PGAS.py:52-55) on v_mfma_f64_16x16x4_f64, whose four-term accumulation is the canonical ascending fma chain: bit-identical to the vector-ALU form.  Default 0: on MI355X the f64 MFMA occupies the SIMD's vector pipeline (no overlap with vector fp64 work, tools/probes/mfma_valu_overlap_probe.hip) and the padded tiles carry 1.5x the flops: 126 against 114 us per step */
#define PGAS_OPT_GRAPH 13 /* 1: pgas_sweep captures its launches (k_init ... k_backtrace, both streams) once in a HIP graph and replays it per sweep on an internal stream -- seed, uniforms, transition parameters, reference trajectory and result all live in device memory the graph's kernels read at execution time; same kernels, same order: identical results.  0: enqueue every launch (what profiled sweeps, the corrected mode and sharded sweeps always do).  Default: off -- on the HIP 7.0 runtime bundled with PyTorch 2.10 the replay measured slower than enqueueing at every size */
#define PGAS_OPT_TRACE_BLOCK_BYTES 12 /* before the first sweep / pgas_shard_setup: keep the traces in row blocks of at most this many bytes (0 = default: one array per trace on an unsharded context, 1 GiB blocks on a shard); small values are a test knob that puts block boundaries inside short sweeps */
#define PGAS_OPT_MNIW_VALU 6 /* 1: pgas_m_mniw_solve factorises column by column on the VALU instead of in MFMA-blocked panels; 2: the two-rows-per-lane kernels of 63 <= M <= 126 at every M (test knobs) */
int pgas_set_option(pgas_ctx* ctx, int32_t option, int64_t value);

/* Free functions of src/Filtering.py on the device.  systematic_SISR(key, w) (:6-37): u = the uniform the reference draws
 * from `key` (:19); weights are passed as log-weights (log 0 = -inf allowed, negative weights have no logarithm: the
 * reference's clip at :23 is the caller's clamp).  reconstruct_trajectory(Particles, ancestry, idx) (:40-55). */
int pgas_systematic_resample(pgas_ctx* ctx, double u, const double* logw_dev, int32_t* idx_dev, void* stream);
/* The same with the uniform u read from device memory at execution time (graph-captured filter steps, include/pgas_marginal.h). */
int pgas_systematic_resample_dev(pgas_ctx* ctx, const double* u_dev, const double* logw_dev, int32_t* idx_dev, void* stream);
int pgas_reconstruct_trajectory(pgas_ctx* ctx, const double* x_dev, const int32_t* anc_dev, int32_t T, int32_t nx, int64_t idx,
                                double* traj_dev, void* stream);

/* ---- particle-sharded sweep.  The reference is single-process; these entry points have no
 * counterpart there.  One context per rank holds N_local = N_global / world particles (N_local a multiple of the
 * segment size).  Per time step there is ONE collective between STEP(t) and GROUPS(t): an all-gather of the segment
 * partials (pgas_shard_buffers: segk_w/segs_w -> segk_g/segs_g); every rank then computes the same group records.
 * Ancestors that live on another rank are read through peer mappings (pgas_shard_set_peer; pgas_ipc_export / pgas_ipc_open
 * carry the mappings between processes).  Results do not depend on `world`. */
#define PGAS_SHARD_INIT 0        /* x_0                      (ref_dev)                         */
#define PGAS_SHARD_PROPAGATE 1   /* time steps [t, t_aux)    (ref_dev)                         */
#define PGAS_SHARD_STEP 2        /* launch t in [1,T]: resample step t-1 (t>1), scan step t (t<T) */
#define PGAS_SHARD_GROUPS 3      /* group records of step t, after the all-gather               */
#define PGAS_SHARD_FINAL_SCAN 4  /* softmax scan of logw_{T-1}; all-gather follows              */
#define PGAS_SHARD_FINAL 5       /* group records + final index (src/PGAS.py:224-225)           */
#define PGAS_SHARD_BACKTRACE 6   /* trajectory (traj_dev), every rank computes the same one     */
int pgas_shard_setup(pgas_ctx* ctx, int32_t rank, int32_t world);
int pgas_shard_buffers(pgas_ctx* ctx, void** out17, int64_t* sizes3);
/* The seven buffers a rank's peers read -- which = 0, 1: the segment cumsums of the two scan buffers; 2..6: the la, h, ln, x and
 * ancestor traces -- block by block: pgas_shard_layout -> info2 = {blocks, rows per block}; pgas_shard_block -> this rank's block;
 * pgas_shard_set_peer_block installs rank `peer`'s block as addressable from this process (every rank, the own one included, must be
 * installed before the first sweep).  All ranks have the same layout (equal N_local and T). */
int pgas_shard_layout(pgas_ctx* ctx, int32_t which, int64_t* info2);
int pgas_shard_block(pgas_ctx* ctx, int32_t which, int32_t blk, void** ptr_dev, int64_t* bytes);
int pgas_shard_set_peer_block(pgas_ctx* ctx, int32_t peer, int32_t which, int32_t blk, const void* ptr_dev);
int pgas_shard_run(pgas_ctx* ctx, int32_t phase, int32_t t, int32_t t_aux, uint64_t seed, const double* ref_dev, double* traj_dev,
                   void* stream);
/* The whole sharded sweep inside the library: pgas_shard_unique_id on one rank (128 bytes, to be broadcast by the host),
 * pgas_shard_comm_init on every rank (ncclCommInitRank with the rank / world of pgas_shard_setup), then pgas_shard_sweep runs
 * the phases above with the per-step all-gather issued as an RCCL call on the same stream -- no host round trip per step.
 * RCCL is bound with dlopen at first use (PGAS_E_STATE if it cannot be loaded).
 * pgas_shard_set_collective replaces the RCCL calls of that loop by a host callback (same loop, same launches): it is how
 * the multi-rank loop is exercised where RCCL is not usable (several ranks sharing one device in the tests).  The callback
 * gets the scan-buffer parity whose segk_w/segs_w must be gathered into segk_g/segs_g (parity < 0: end-of-sweep barrier)
 * and the stream the loop runs on; it must leave the gathered arrays visible to work enqueued on that stream afterwards. */
typedef int (*pgas_allgather_fn)(void* user, int32_t parity, void* stream);
int pgas_shard_unique_id(void* id128);
int pgas_shard_comm_init(pgas_ctx* ctx, const void* id128);
int pgas_shard_set_collective(pgas_ctx* ctx, pgas_allgather_fn fn, void* user);
int pgas_shard_sweep(pgas_ctx* ctx, uint64_t seed, const double* ref_dev, double* traj_dev, int32_t propagate_chunk, void* stream);
/* Measurement aid: issue the step's RCCL all-gather `reps` times on `stream` (between sweeps only). */
int pgas_shard_probe_collective(pgas_ctx* ctx, int32_t reps, void* stream);
/* Version of the HIP runtime this library is bound to in the running process (hipRuntimeGetVersion: 7.2.x = 702xxxxx), -1 on failure
 * (recorded by bench.py; the HIP 7.0 runtime bundled with PyTorch 2.10+rocm7.0 never returns from hipIpcOpenMemHandle for an
 * allocation of 2 GiB or more, which is why a shard's traces are row blocks of at most 1 GiB). */
int32_t pgas_hip_runtime_version(void);

/* HIP IPC plumbing for peers in OTHER processes: a 64-byte handle of block `blk` of this rank's buffer `which` (as in
 * pgas_shard_layout), and the mapping of a peer's handle into this process (xGMI peer access is enabled on first use). */
int pgas_ipc_export(pgas_ctx* ctx, int32_t which, int32_t blk, void* handle64);
int pgas_ipc_open(pgas_ctx* ctx, const void* handle64, void** ptr);

/* Test hook: the arithmetic primitives shared with the CPU oracle (include/pgas_detmath.h, include/pgas_canon.h) evaluated ON THE
 * DEVICE, element by element, so that the shared header is checked on gfx950 directly and not only through whole-step parity:
 * which = 0 exp(x), 1 log(x), 2 sin(pi x) -> out0 / cos(pi x) -> out1, 3 Philox4x32-10 (w: 6 words per element = counter[4], key[2];
 * outw: 4 words), 4 pgas_seg_ref(x), 5 pgas_seg_arg(x, y), 6 pgas_lvl_scale(x, y), 7 the Box-Muller pair of the Philox block of w. */
int pgas_detmath_eval(int32_t device, int32_t which, const double* x_dev, const double* y_dev, const uint32_t* w_dev, int64_t n,
                      double* out0_dev, double* out1_dev, uint32_t* outw_dev, void* stream);

/* Sufficient statistics of PGAS.sample_params (src/PGAS.py:294-303, BI:53-61) without the prior:
 * traj_dev (T,nx) -> T0 (M,nx), T1 (M,M) [fp64 MFMA SYRK], T2 (nx,nx); T3 = T-1.
 * Pairs traj[:-1] with inputs[:-1] (quirk Q3). */
int pgas_suffstats(pgas_ctx* ctx, const double* traj_dev, double* T0_dev, double* T1_dev, double* T2_dev, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PGAS_HIP_H */
