// pgas_api.hip -- C ABI of libpgas_hip.so (include/pgas_hip.h) over the kernels of
// pgas_kernels.hip.h.  Host side only: argument checking, device tables, launch sequencing.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/pgas_hip.h"
#include "pgas_kernels.hip.h"
#include "pgas_resample.hip.h"
#include "pgas_suffstats.hip.h"
#include "pgas_marginal.hip.h"
#include "../../include/pgas_marginal.h"

#ifndef PG_W3
#define PG_W3 2    // waves per SIMD the 3-D k_propagate instantiations are compiled for
#endif
#ifndef PG_W28
#define PG_W28 2
#endif
#ifndef PG_WMX
#define PG_WMX 2    // waves per SIMD the matrix-core k_propagate is compiled for
#endif
#ifndef PG_PPT3
#define PG_PPT3 8   // particles per thread of the 3-D, nx = 2 variants (EMPS / Vehicle M = 729)
#endif
#ifndef PG_PPT27
#define PG_PPT27 4   // particles per thread of that variant = one segment per workgroup
#endif
#ifndef PG_P27
#define PG_P27 4   // particles per basis pass of the SingleMassOscillator-shaped FAST variant.  Round 3, interleaved A/B at N = 2^20 (ms per
                   // sweep; the round-2 kernels had preferred PPT 8 / P 2): PPT 8 / P 2 68.1-69.4, 4 / 2 65.0, 4 / 1 70.4, 8 / 4 67.5, 8 / 8 66.9,
                   // 4 / 4 63.0-64.9 -- one pass over the thread's four particles: every coefficient (a scalar operand) feeds four FMAs, no
                   // next-group prefetch registers (123 VGPRs instead of 141-151: four waves per SIMD fit), 1024 workgroups
#endif
#ifndef PG_P3
#define PG_P3 4   // particles per basis pass of the 3-D, nx = 2 variants: every coefficient read from LDS feeds four FMAs (measured EMPS-729 at N = 2^20: P = 1 208, P = 2 158, P = 4 147 us per step)
#endif
namespace {

thread_local std::string g_create_error;

// RCCL is bound at run time (dlopen), so the library loads on machines without it and shares whichever RCCL the process
// already has (PyTorch ships its own copy): only the particle-sharded sweep needs it.
struct RcclApi {
    ncclResult_t (*get_unique_id)(ncclUniqueId*) = nullptr;
    ncclResult_t (*init_rank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*all_gather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*group_start)() = nullptr;
    ncclResult_t (*group_end)() = nullptr;
    ncclResult_t (*destroy)(ncclComm_t) = nullptr;
    const char* (*error_string)(ncclResult_t) = nullptr;
    bool ok = false;
};
RcclApi& rccl() {
    static RcclApi api = [] {
        RcclApi a;
        void* h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) return a;
        a.get_unique_id = (decltype(a.get_unique_id))dlsym(h, "ncclGetUniqueId");
        a.init_rank = (decltype(a.init_rank))dlsym(h, "ncclCommInitRank");
        a.all_gather = (decltype(a.all_gather))dlsym(h, "ncclAllGather");
        a.group_start = (decltype(a.group_start))dlsym(h, "ncclGroupStart");
        a.group_end = (decltype(a.group_end))dlsym(h, "ncclGroupEnd");
        a.destroy = (decltype(a.destroy))dlsym(h, "ncclCommDestroy");
        a.error_string = (decltype(a.error_string))dlsym(h, "ncclGetErrorString");
        a.ok = a.get_unique_id && a.init_rank && a.all_gather && a.group_start && a.group_end && a.destroy;
        return a;
    }();
    return api;
}

// Every entry point runs on its context's device and puts the caller's current device back on return (all paths, errors included):
// a process that drives several contexts, or keeps torch tensors on another device, must not see its current device change.
struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) == hipSuccess && prev != dev) switched = hipSetDevice(dev) == hipSuccess;
        else if (prev < 0) (void)hipSetDevice(dev);
    }
    ~DeviceGuard() {
        if (switched) (void)hipSetDevice(prev);
    }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};

typedef void (*front_fn)(DevModel, const TransParams*, int, uint64_t, const double*, const double*, const double*, int, double*, ScanBufs);
typedef void (*backc_fn)(DevModel, const TransParams*, int, uint64_t, double, const double*, const double*, ScanBufs, Peers, int32_t*, double*, double*);
typedef void (*prop_fn)(DevModel, const TransParams*, const double*, const SweepParams*, int, int, const double*, double*, const double*, double*, double*, double*);
typedef void (*propmx_fn)(DevModel, const TransParams*, const double*, const SweepParams*, int, int, const double*, double*, const double*, double*, double*, double*,
                          const MxInfo*, const double*);
typedef void (*aux_fn)(DevModel, const TransParams*, int, const double*, double*);
typedef void (*back_fn)(DevModel, int, double, const double*, ScanBufs, Peers, int32_t*, double*);
typedef void (*init_fn)(DevModel, uint64_t, const SweepParams*, const double*, const double*, double*);
typedef void (*basis_fn)(DevModel, const int32_t*, const double*, int64_t, int, double*);
typedef void (*duo_fn)(DevModel, const TransParams*, const double*, const SweepParams*, const double*, const double*, const double*, const double*, double*, int32_t*,
                       double*, double*, UpperHdr*, double*, const double*, DuoShared*);
typedef void (*small_fn)(DevModel, const TransParams*, const double*, const SweepParams*, const double*, const double*, const double*, const double*, double*, int32_t*,
                         double*, double*, UpperHdr*, double*, const double*);

struct Variant {
    front_fn front;
    prop_fn prop;
    prop_fn prop_one;   // instantiation for launches of exactly one time step (nullptr: use prop)
    aux_fn aux;
    int P, W;   // particles per basis pass, waves per SIMD the k_propagate instantiation is built for
    int PPT;    // particles per thread of k_propagate: its grid is ceil(nseg / (PPT / 4))
    small_fn small[3];   // the whole sweep in one workgroup (N <= 256, 512, 1024: one, two, four particles per thread)
    duo_fn duo[3];       // ... on two workgroups: propagation ahead, weight recursion behind (the default)
};

template <int NX, int D, int JIN, int P, int W, int J0T = 0, int PPT = PG_PPT>
Variant make_variant() {
    const prop_fn one = k_propagate<NX, D, JIN, P, W, J0T, PPT, true>;
    return Variant{k_front<NX, D, JIN, P>, k_propagate<NX, D, JIN, P, W, J0T, PPT>, one, k_aux<NX, D, JIN, P>, P, W, PPT,
                   {k_sweep_small<NX, D, JIN, J0T, 1>, k_sweep_small<NX, D, JIN, J0T, 2>, k_sweep_small<NX, D, JIN, J0T, 4>},
                   {k_sweep_duo<NX, D, JIN, J0T, 1>, k_sweep_duo<NX, D, JIN, J0T, 2>, k_sweep_duo<NX, D, JIN, J0T, 4>}};
}

// (nx, D, padded innermost extent) -> kernel instantiation <NX, D, JIN, P particles per basis pass, W waves/SIMD>
// `fast`: the model has the shape the FAST instantiations are compiled for (sel[d] == d, jstep == j0 in every dimension);
// J0 = frequencies of the outermost dimension.  SingleMassOscillator: 7 x 7 (the 729-function bases of EMPS / Vehicle select a ball of
// the 11 x 11 x 11 frequency grid and take the generic 3-D instantiation, which skips the empty part of every row).
bool pick_variant(int nx, int D, int jin_needed, bool fast, int J0, Variant* v, int* JP) {
    if (fast && nx == 2 && D == 2 && jin_needed == 7 && J0 == 7) {
        *JP = 7;
        *v = make_variant<2, 2, 7, PG_P27, PG_W28, 7, PG_PPT27>();
        return true;
    }
    if (D == 1) {
        *JP = 1;
        *v = nx == 1 ? make_variant<1, 1, 1, 4, 2>() : make_variant<2, 1, 1, 4, 2>();
        return true;
    }
    const int jp = jin_needed <= 8 ? 8 : jin_needed <= 12 ? 12 : jin_needed <= 16 ? 16 : 0;
    if (!jp) return false;
    *JP = jp;
    if (nx == 1) {
        if (D == 2) *v = jp == 8 ? make_variant<1, 2, 8, 2, 2>() : jp == 12 ? make_variant<1, 2, 12, 2, 2>() : make_variant<1, 2, 16, 2, 2>();
        else *v = jp == 8 ? make_variant<1, 3, 8, 2, 2>() : jp == 12 ? make_variant<1, 3, 12, 2, 2>() : make_variant<1, 3, 16, 2, 2>();
    } else {
        if (D == 2) *v = jp == 8 ? make_variant<2, 2, 8, 2, PG_W28>() : jp == 12 ? make_variant<2, 2, 12, 2, 2>() : make_variant<2, 2, 16, 2, 2>();
        else *v = jp == 8 ? make_variant<2, 3, 8, PG_P3, PG_W3, 0, PG_PPT3>() : jp == 12 ? make_variant<2, 3, 12, PG_P3, PG_W3, 0, PG_PPT3>() : make_variant<2, 3, 16, PG_P3, PG_W3, 0, PG_PPT3>();
    }
    return true;
}

// One trace buffer (T or T-1 rows of equal size) as row blocks: one allocation in the contiguous layout (unsharded contexts: the
// caller gets plain (T, N, ...) arrays from pgas_get_traces), power-of-two runs of rows of at most `block_bytes` each otherwise
// (sharded contexts: every block stays below the 2 GiB above which hipIpcOpenMemHandle of PyTorch's bundled runtime hangs).
struct RowStore {
    std::vector<void*> own;     // allocations
    std::vector<char*> blk;     // base of block b = rows [b << shift, (b + 1) << shift)
    size_t row_bytes = 0;
    int rows = 0, shift = 30, nblk = 0;
    bool contiguous = true;
    char* row(int t) const { return blk[(size_t)t >> shift] + (size_t)(t & ((1 << shift) - 1)) * row_bytes; }
    int rows_in_block(int b) const { const int r0 = b << shift, r1 = r0 + (1 << shift); return (r1 < rows ? r1 : rows) - r0; }
    void release() {
        for (void* p : own) hipFree(p);
        own.clear(); blk.clear(); nblk = 0; rows = 0;
    }
};
#define PG_TRACE_BLOCK_BYTES ((size_t)1 << 30)
// PGAS_OPT_GRAPH automatic: replay the captured sweep up to this many particles.  Measured (PyTorch 2.10's bundled HIP 7.0 runtime,
// T = 2000): at N = 2^20 the replay is SLOWER than enqueueing -- hipGraphLaunch itself blocks the host for 53 ms (eager enqueue: 45 ms)
// and the replayed sweep takes 92 ms on the device against 66-70 ms (the two pipelines no longer overlap as well) -- so large sweeps
// stay on the eager path; DESIGN.md section 8 has the small-N figures.
#define PG_GRAPH_AUTO_N 0

}  // namespace

struct pgas_ctx {
    DevModel md{};
    int device = 0;
    int keep_logw = 0;
    Variant var{};
    back_fn back = nullptr;
    backc_fn back_corrected = nullptr;
    int corrected = 0;          // PGAS_OPT_RESAMPLE_BEFORE_PROPAGATE: propagate from the resampled ancestors (quirk Q1 removed)
    double* aux_buf = nullptr;  // (N, nx) transition means of the current step, corrected mode only
    ncclComm_t comm = nullptr;  // RCCL communicator of the particle-sharded sweep (pgas_shard_comm_init)
    int32_t* d_sync = nullptr;  // one word all-gathered at the end of a sharded sweep: orders peer reads before the next sweep's writes
    int32_t* d_fail = nullptr;  // failure counter of pgas_m_mniw_solve
    int mniw_valu = 0;          // PGAS_OPT_MNIW_VALU: 1 = column-by-column VALU factorisation instead of the MFMA-blocked one
    double* ws_partial = nullptr;  // per-chunk partial sums of pgas_m_weighted_stats
    size_t ws_bytes = 0;
    init_fn init = nullptr;
    basis_fn basis = nullptr;
    // device tables
    double* d_y = nullptr;
    double* d_u = nullptr;
    int32_t* d_idx = nullptr;
    int32_t* d_pos = nullptr;
    uint64_t* d_qdesc = nullptr;   // 3-D bases: packed leading non-zero counts per (a, b) row of the frequency grid
    double* d_m0L0 = nullptr;
    double* d_ref = nullptr;   // nx doubles: ref_t of pgas_step / ref0 of pgas_init_state
    double* d_G = nullptr;
    int64_t gtotal = 0;        // doubles in d_G
    TransParams tp{};           // host mirror of what pgas_set_params was given (pgas_set_params_dev: only G is meaningful)
    TransParams* d_tp = nullptr;   // THE transition parameters every kernel reads (k_pack writes them)
    SweepParams* d_sp = nullptr;   // per-sweep scalars (k_sweep_begin)
    double* d_ures = nullptr;      // (T + 1) resampling uniforms of the running sweep
    double* d_uanc = nullptr;      // (T + 1) ancestor uniforms
    uint32_t epoch = 0;            // sweeps started on this context
    // captured sweep (PGAS_OPT_GRAPH): k_init ... k_backtrace of pgas_sweep as one HIP graph, replayed per sweep; the reference
    // trajectory and the result go through library-owned buffers because the caller's pointers change from call to call
    int use_graph = -1;            // PGAS_OPT_GRAPH: 1 on, 0 off, -1 automatic (see pgas_sweep)
    hipGraph_t graph = nullptr;
    hipGraphExec_t graph_exec = nullptr;
    int graph_key[6] = {-1, -1, -1, -1, -1, -1};   // launch configuration the graph was captured with
    int last_graph = 0;            // 1: the last pgas_sweep replayed the captured graph
    // 3-D, nx = 2, innermost extent 12 (EMPS / Vehicle M = 729): k_propagate's contraction on the f64 matrix cores (PGAS_OPT_MFMA_PROPAGATE)
    MxInfo* d_mxi = nullptr;       // tile descriptor + gather indices of the operand image
    double* d_gimg = nullptr;      // the image (k_pack_mx, per parameter set)
    int mx_slots = 0;
    int use_mx = 0;                // measured slower than the vector form (126 vs 114 us per step): opt-in
    int use_small = 1;             // PGAS_OPT_SMALL_SWEEP: contexts of at most one segment run the whole sweep in one workgroup (k_sweep_small)
    int last_small = 0;            // 1: the last pgas_sweep did
    double* d_znoise = nullptr;    // (T, N, 2) propagation noise of a single-workgroup sweep (k_small_noise)
    DuoShared* d_duo = nullptr;    // ring and counters between the two workgroups of k_sweep_duo
    int graph_failed = 0;          // capture or instantiation failed once: stay on the eager path
    hipStream_t sG = nullptr;      // the stream captured sweeps are recorded on and replayed on (the caller's may be the legacy default
                                   // stream, which cannot be captured); ordered against the caller's stream with ev_g0 / ev_g1
    hipEvent_t ev_g0 = nullptr, ev_g1 = nullptr;
    double* d_refbuf = nullptr;    // (T, nx)
    double* d_trajbuf = nullptr;   // (T, nx)
    bool have_params = false;
    ScanBufs sb[2]{};
    // traces: rs[PG_RB_X] (T rows of (N, nx)), rs[PG_RB_ANC] (T-1 rows of N int32), and the hand-off rows k_propagate writes for the
    // weight recursion: rs[PG_RB_LA] log p(y_t | aux_t), rs[PG_RB_H] log N(ref_t; aux_t, S), rs[PG_RB_LN] log p(y_t | x_t), T rows of nseg*SEG each
    RowStore rs[PG_RB_NKIND];
    bool have_traces = false;
    size_t trace_block_bytes = 0;   // PGAS_OPT_TRACE_BLOCK_BYTES: 0 = contiguous (unsharded) / PG_TRACE_BLOCK_BYTES (sharded)
    std::vector<const char*> peer_blk[PG_MAX_RANKS][PG_RB_NKIND];   // every rank's block bases as mapped into this process (own rank: rs[].blk)
    const void** d_bt = nullptr;    // device copy of the x / ancestor block table for k_backtrace
    size_t d_bt_entries = 0;
    bool bt_dirty = true;
    double* logw_last = nullptr;
    double* logw_trace = nullptr;
    double* laux_own[2] = {nullptr, nullptr};  // laux of the two step-API scan buffers
    int prop_chunk = 0;         // time steps per k_propagate launch (0 = whole sweep)
    Peers peers{};              // world == 1 unless pgas_shard_setup was called
    const uint64_t* peer_c1[2][PG_MAX_RANKS] = {};  // per scan-buffer parity
    const uint64_t* peer_c2[2][PG_MAX_RANKS] = {};
    bool sharded = false;
    int rank = 0, world = 1;
    double* segk_g[2] = {nullptr, nullptr};   // gathered partials (sharded mode): (world, 2, nsegp) per scan buffer
    uint64_t* segs_g[2] = {nullptr, nullptr};
    pgas_allgather_fn ag_cb = nullptr;        // host-staged all-gather installed by pgas_shard_set_collective (tests); NULL = RCCL
    void* ag_user = nullptr;
    int last_chunk = 0;         // time steps per k_propagate launch of the last sweep
    int var_P = 0;              // particles per basis pass of the k_propagate variant
    unsigned launch_tag = 0;    // unique id per k_step launch (hand-off word tag)
    int force_slow = 0;         // 1: never let k_step scan the groups itself (test hook for the k_groups path taken when N > 2^20 per device)
    int tail_groups = 0;        // PGAS_OPT_TAIL_GROUPS: 1 = group scans in k_step's tail instead of k_groups launches (single device; slower, kept as an experiment)
    int ev_stride = 8;          // PGAS_OPT_EVENT_STRIDE: k_propagate launches per event that gates the weight recursion
    int local_groups = 0;       // PGAS_OPT_LOCAL_GROUPS: 1 = k_step<LOCAL> where it applies
    int overlap = 1;            // 1: run the weight recursion on an internal stream concurrently with k_propagate
    int prop_lds = 0;   // dynamic LDS reserved per k_propagate workgroup when overlapping: caps it at two workgroups per CU so
                                // that two k_step workgroups always fit beside them
    hipStream_t sB = nullptr;   // internal stream of the weight recursion
    std::vector<hipEvent_t> ev_chunk;  // "k_propagate chunk c done" events
    hipEvent_t ev_start = nullptr, ev_done = nullptr;
    // suff-stat scratch
    double* d_phi = nullptr;
    double* d_syrk_ws = nullptr;   // split-K partial slabs of Z^T Z (pgas_suffstats)
    size_t syrk_ws_bytes = 0;
    int syrk_splits = 0;           // 0 = automatic
    bool peer_access_tried = false;
    const uint32_t* t_dev = nullptr;   // pgas_m_set_time_source: time index of the marginalised family's random-number kernels, device-resident
    int max_lead = 0;              // PGAS_OPT_MAX_LEAD
    std::vector<hipEvent_t> ev_bdone;
    // optional per-launch timing of the dominant kernel (pgas_set_profiling)
    int profiling = 0;
    int prof_stride = 16;       // every prof_stride-th launch carries start/stop events (hipExtLaunchKernelGGL: the dispatch's own timestamps)
    std::vector<hipEvent_t> ev;      // pairs (start, stop) around each k_step launch of the last sweep
    std::vector<hipEvent_t> evp;     // pairs (start, stop) around each k_propagate launch of the last sweep
    int evp_used = 0;
    int ev_used = 0;
    std::string err;
};

#define FAIL(ctx, code, ...)                                  \
    do {                                                      \
        char buf_[512];                                       \
        snprintf(buf_, sizeof buf_, __VA_ARGS__);             \
        (ctx)->err = buf_;                                    \
        return (code);                                        \
    } while (0)

#define HIPCHK(ctx, call)                                                                       \
    do {                                                                                        \
        hipError_t e_ = (call);                                                                 \
        if (e_ != hipSuccess) FAIL(ctx, e_ == hipErrorOutOfMemory ? PGAS_E_NOMEM : PGAS_E_HIP,  \
                                   "%s failed: %s", #call, hipGetErrorString(e_));              \
    } while (0)

#define KCHK(ctx, name)                                                                        \
    do {                                                                                        \
        hipError_t e_ = hipGetLastError();                                                      \
        if (e_ != hipSuccess) FAIL(ctx, PGAS_E_HIP, "launch of %s failed: %s", name, hipGetErrorString(e_)); \
    } while (0)

static int alloc_scanbufs(pgas_ctx* c, ScanBufs* sb) {
    const size_t np = (size_t)c->md.nseg * PGAS_SEG;
    const int nsegp = (c->md.nseg + 63) / 64 * 64;
    sb->nsegp = nsegp;
    HIPCHK(c, hipMalloc(&sb->laux, np * sizeof(double)));
    HIPCHK(c, hipMalloc(&sb->c1, np * sizeof(uint64_t)));
    HIPCHK(c, hipMalloc(&sb->c2, np * sizeof(uint64_t)));
    HIPCHK(c, hipMalloc(&sb->segk_w, 2 * nsegp * sizeof(double)));
    HIPCHK(c, hipMalloc(&sb->segs_w, 2 * nsegp * sizeof(uint64_t)));
    HIPCHK(c, hipMalloc(&sb->tab_e, 2 * nsegp * sizeof(double)));
    HIPCHK(c, hipMalloc(&sb->tab_sc, 2 * nsegp * sizeof(double)));
    HIPCHK(c, hipMalloc(&sb->tab_m, 2 * nsegp * sizeof(double)));
    HIPCHK(c, hipMalloc(&sb->grp_K, 2 * PG_MAX_GRP * sizeof(double)));
    HIPCHK(c, hipMalloc(&sb->grp_T, 2 * PG_MAX_GRP * sizeof(double)));
    HIPCHK(c, hipMalloc(&sb->grp_cnt, PG_MAX_GRP * sizeof(unsigned)));
    HIPCHK(c, hipMemset(sb->grp_cnt, 0, PG_MAX_GRP * sizeof(unsigned)));
    HIPCHK(c, hipMalloc(&sb->hdr, sizeof(UpperHdr)));
    HIPCHK(c, hipMemset(sb->hdr, 0, sizeof(UpperHdr)));
    sb->segk = sb->segk_w;
    sb->segs = sb->segs_w;
    sb->nseg_l = 0x40000000;
    sb->rank_stride = 0;
    sb->nsegp_g = nsegp;
    return PGAS_OK;
}
static void free_scanbufs(ScanBufs* sb) {
    hipFree(sb->laux); hipFree(sb->c1); hipFree(sb->c2); hipFree(sb->segk_w); hipFree(sb->segs_w);  // segk/segs alias these or the gathered arrays
    hipFree(sb->tab_e); hipFree(sb->tab_sc); hipFree(sb->tab_m); hipFree(sb->grp_K); hipFree(sb->grp_T); hipFree(sb->grp_cnt); hipFree(sb->hdr);
    *sb = ScanBufs{};
}

extern "C" {

int32_t pgas_segment_size(void) { return PGAS_SEG; }

const char* pgas_last_error(const pgas_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

static int create_impl(const pgas_model_desc* d, pgas_ctx* c) {
    if (!d->idx || !d->sel || !d->alpha || !d->beta || !d->H || !d->LRinv || !d->m0 || !d->L0 || !d->y)
        FAIL(c, PGAS_E_ARG, "pgas_create: NULL table pointer");
    if (d->N < 1 || d->T < 1 || d->M < 1) FAIL(c, PGAS_E_ARG, "pgas_create: N, T, M must be >= 1");
    if (d->nx < 1 || d->nx > 2) FAIL(c, PGAS_E_ARG, "pgas_create: nx = %d not supported (compiled for nx in {1,2})", d->nx);
    if (d->ny < 1 || d->ny > PGAS_MAX_NY) FAIL(c, PGAS_E_ARG, "pgas_create: ny = %d not supported (1..%d)", d->ny, PGAS_MAX_NY);
    if (d->nu < 0 || d->nu > PGAS_MAX_NU) FAIL(c, PGAS_E_ARG, "pgas_create: nu = %d not supported (0..%d)", d->nu, PGAS_MAX_NU);
    if (d->D < 1 || d->D > PGAS_MAX_D) FAIL(c, PGAS_E_ARG, "pgas_create: D = %d not supported (1..%d)", d->D, PGAS_MAX_D);
    if (d->nu > 0 && !d->u) FAIL(c, PGAS_E_ARG, "pgas_create: nu > 0 but u == NULL");
    const int64_t nseg64 = ((int64_t)d->N + PGAS_SEG - 1) / PGAS_SEG;
    if (nseg64 > PG_MAX_NSEG) FAIL(c, PGAS_E_ARG, "pgas_create: N = %d exceeds %d particles per device", d->N, PG_MAX_NSEG * PGAS_SEG);

    DevModel& md = c->md;
    md.N = d->N; md.T = d->T; md.nx = d->nx; md.ny = d->ny; md.nu = d->nu; md.D = d->D; md.M = d->M;
    md.nseg = (int)nseg64;
    md.p0 = 0; md.Ng = d->N; md.nseg_g = md.nseg;
    c->peers = Peers{}; c->peers.world = 1; c->peers.nseg_l = md.nseg; c->peers.Nl = d->N;
    md.nrm = d->nrm; md.cR = d->cR;
    for (int k = 0; k < d->D; ++k) {
        if (d->sel[k] < 0 || d->sel[k] >= d->nx + d->nu) FAIL(c, PGAS_E_ARG, "pgas_create: sel[%d] = %d out of range", k, d->sel[k]);
        md.sel[k] = d->sel[k]; md.alpha[k] = d->alpha[k]; md.beta[k] = d->beta[k];
        // frequency progression of dimension k: j0, j0 + step, ...   (src/BasisFunctions.py:24-25)
        int lo = d->idx[k], hi = d->idx[k], second = 0;
        for (int m = 0; m < d->M; ++m) {
            const int j = d->idx[m * d->D + k];
            if (j < 1) FAIL(c, PGAS_E_ARG, "pgas_create: idx[%d][%d] = %d must be >= 1", m, k, j);
            lo = j < lo ? j : lo; hi = j > hi ? j : hi;
        }
        for (int m = 0; m < d->M; ++m) {
            const int j = d->idx[m * d->D + k];
            if (j > lo && (second == 0 || j < second)) second = j;
        }
        md.j0[k] = lo; md.jstep[k] = second ? second - lo : 1;
        for (int m = 0; m < d->M; ++m)
            if ((d->idx[m * d->D + k] - lo) % md.jstep[k]) FAIL(c, PGAS_E_ARG, "pgas_create: idx column %d is not an arithmetic progression", k);
        md.J[k] = (hi - lo) / md.jstep[k] + 1;
        if (md.J[k] > PGAS_MAX_J) FAIL(c, PGAS_E_ARG, "pgas_create: %d frequencies in dimension %d (max %d)", md.J[k], k, PGAS_MAX_J);
    }
    for (int j = 0; j < d->ny; ++j)
        for (int k = 0; k < d->nx; ++k) md.H[j * d->nx + k] = d->H[j * d->nx + k];
    for (int j = 0; j < d->ny * d->ny; ++j) md.LRinv[j] = d->LRinv[j];
    bool fast = !d->no_fast_variant;
    for (int k = 0; k < d->D; ++k) fast = fast && md.sel[k] == k && md.jstep[k] == md.j0[k];
    if (!pick_variant(d->nx, d->D, md.J[d->D - 1], fast, md.J[0], &c->var, &md.JP))
        FAIL(c, PGAS_E_ARG, "pgas_create: innermost basis dimension has %d frequencies (compiled up to 16)", md.J[d->D - 1]);
    c->var_P = c->var.P;
    c->back = d->nx == 1 ? k_back<1> : k_back<2>;
    c->back_corrected = d->nx == 1 ? k_back_corrected<1> : k_back_corrected<2>;
    c->init = d->nx == 1 ? k_init<1> : k_init<2>;
    c->basis = d->nx == 1 ? k_basis_eval<1> : k_basis_eval<2>;
    c->keep_logw = d->keep_logw_trace;

    c->device = d->device;
    // grid positions of the basis functions, innermost dimension padded to JP
    std::vector<int32_t> pos(d->M);
    int64_t gsize = 1;
    for (int k = 0; k < d->D; ++k) gsize *= (k == d->D - 1 && d->D > 1) ? md.JP : md.J[k];
    for (int m = 0; m < d->M; ++m) {
        int64_t p = 0;
        for (int k = 0; k < d->D; ++k) {
            const int ext = (k == d->D - 1 && d->D > 1) ? md.JP : md.J[k];
            p = p * ext + (d->idx[m * d->D + k] - md.j0[k]) / md.jstep[k];
        }
        pos[m] = (int32_t)p;
    }
    if (d->D == 3 && md.J[1] <= 12) {
        // row descriptors of the frequency grid: 5 bits per (a, b) = leading innermost frequencies in use (0..JP <= 16)
        std::vector<uint64_t> qd((size_t)md.J[0], 0ull);
        std::vector<int> ql((size_t)md.J[0] * md.J[1], 0);
        for (int m = 0; m < d->M; ++m) {
            const int a = (d->idx[m * 3 + 0] - md.j0[0]) / md.jstep[0], b = (d->idx[m * 3 + 1] - md.j0[1]) / md.jstep[1];
            const int q = (d->idx[m * 3 + 2] - md.j0[2]) / md.jstep[2] + 1;
            if (q > ql[(size_t)a * md.J[1] + b]) ql[(size_t)a * md.J[1] + b] = q;
        }
        for (int a = 0; a < md.J[0]; ++a)
            for (int b = 0; b < md.J[1]; ++b) qd[a] |= (uint64_t)ql[(size_t)a * md.J[1] + b] << (5 * b);
        HIPCHK(c, hipMalloc(&c->d_qdesc, qd.size() * sizeof(uint64_t)));
        HIPCHK(c, hipMemcpy(c->d_qdesc, qd.data(), qd.size() * sizeof(uint64_t), hipMemcpyHostToDevice));
        md.qdesc = c->d_qdesc;
        if (d->nx == 2 && md.JP == 12 && md.J[0] <= 12) {
            // matrix-core form of the contraction (eval_mean_mx): pair ci = outermost frequencies (2 ci, 2 ci + 1), tile tau = b in
            // [4 tau, 4 tau + 4); a tile runs as many K steps (of 4 innermost frequencies) as its longest row needs
            // The kernel is compiled for ONE descriptor, PG_MX_DESC_BALL729 (every 729-function basis of the reference); a model whose tiles
            // need no more K steps than that anywhere runs on it (extra K steps multiply zeros), any other keeps the vector form.
            std::vector<MxInfo> mi(1);
            MxInfo& m = mi[0];
            m.desc = PG_MX_DESC_BALL729; m.nslots = 0; m.pad = 0;
            const int J0 = md.J[0], J1 = md.J[1];
            bool fits = true;
            for (int ci = 0; ci < 6; ++ci) {
                int nks[3] = {0, 0, 0};
                for (int tau = 0; tau < 3; ++tau) {
                    int need = 0;
                    for (int a = 2 * ci; a < 2 * ci + 2 && a < J0; ++a)
                        for (int b = 4 * tau; b < 4 * tau + 4 && b < J1; ++b) need = std::max(need, (ql[(size_t)a * J1 + b] + 3) / 4);
                    nks[tau] = mx_nk(PG_MX_DESC_BALL729, 3 * ci + tau);
                    fits = fits && need <= nks[tau];
                }
                for (int tau = 0; tau < 3; ++tau)       // the order eval_mean_mx issues its MFMAs in
                    for (int ks = 0; ks < nks[tau]; ++ks) {
                        for (int lane = 0; lane < 64; ++lane) {
                            const int i = lane % 16, q = 4 * ks + lane / 16, g = i % 4, b = 4 * tau + i / 4, a = 2 * ci + g / 2, k = g % 2;
                            m.idx[m.nslots * 64 + lane] = (a < J0 && b < J1) ? (int32_t)((((size_t)a * J1 + b) * md.JP + q) * 2 + k) : -1;
                        }
                        ++m.nslots;
                    }
            }
            if (fits) {
                c->mx_slots = m.nslots;
                HIPCHK(c, hipMalloc(&c->d_mxi, sizeof(MxInfo)));
                HIPCHK(c, hipMemcpy(c->d_mxi, &m, sizeof(MxInfo), hipMemcpyHostToDevice));
                HIPCHK(c, hipMalloc(&c->d_gimg, (size_t)m.nslots * 64 * sizeof(double)));
            }
        }
    }
    c->gtotal = gsize * d->nx;
    HIPCHK(c, hipMalloc(&c->d_G, c->gtotal * sizeof(double)));
    HIPCHK(c, hipMalloc(&c->d_pos, d->M * sizeof(int32_t)));
    HIPCHK(c, hipMemcpy(c->d_pos, pos.data(), d->M * sizeof(int32_t), hipMemcpyHostToDevice));
    HIPCHK(c, hipMalloc(&c->d_idx, (size_t)d->M * d->D * sizeof(int32_t)));
    HIPCHK(c, hipMemcpy(c->d_idx, d->idx, (size_t)d->M * d->D * sizeof(int32_t), hipMemcpyHostToDevice));
    HIPCHK(c, hipMalloc(&c->d_y, (size_t)d->T * d->ny * sizeof(double)));
    HIPCHK(c, hipMemcpy(c->d_y, d->y, (size_t)d->T * d->ny * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(c, hipMalloc(&c->d_u, ((size_t)d->T * d->nu + 1) * sizeof(double)));
    if (d->nu) HIPCHK(c, hipMemcpy(c->d_u, d->u, (size_t)d->T * d->nu * sizeof(double), hipMemcpyHostToDevice));
    md.y = c->d_y; md.u = c->d_u;
    std::vector<double> m0L0(d->nx + d->nx * d->nx);
    for (int k = 0; k < d->nx; ++k) m0L0[k] = d->m0[k];
    for (int k = 0; k < d->nx * d->nx; ++k) m0L0[d->nx + k] = d->L0[k];
    HIPCHK(c, hipMalloc(&c->d_m0L0, m0L0.size() * sizeof(double)));
    HIPCHK(c, hipMemcpy(c->d_m0L0, m0L0.data(), m0L0.size() * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(c, hipMalloc(&c->d_ref, 4 * sizeof(double)));
    HIPCHK(c, hipMalloc(&c->d_tp, sizeof(TransParams)));
    HIPCHK(c, hipMalloc(&c->d_sp, sizeof(SweepParams)));
    HIPCHK(c, hipMemset(c->d_sp, 0, sizeof(SweepParams)));
    HIPCHK(c, hipMalloc(&c->d_ures, ((size_t)d->T + 1) * sizeof(double)));
    HIPCHK(c, hipMalloc(&c->d_uanc, ((size_t)d->T + 1) * sizeof(double)));
    HIPCHK(c, hipMalloc(&c->d_refbuf, (size_t)d->T * d->nx * sizeof(double)));
    HIPCHK(c, hipMalloc(&c->d_trajbuf, (size_t)d->T * d->nx * sizeof(double)));
    { const char* e = getenv("PGAS_GRAPH"); if (e && (e[0] == '0' || e[0] == '1')) c->use_graph = e[0] - '0'; }   // development knob
    { const char* e = getenv("PGAS_MFMA_PROPAGATE"); if (e && (e[0] == '0' || e[0] == '1')) c->use_mx = e[0] - '0'; }   // development knob (A/B runs)
    for (int i = 0; i < 2; ++i) {
        int rc = alloc_scanbufs(c, &c->sb[i]);
        if (rc) return rc;
    }
    return PGAS_OK;
}

int pgas_create(const pgas_model_desc* desc, pgas_ctx** out) {
    if (!desc || !out) { g_create_error = "pgas_create: NULL argument"; return PGAS_E_ARG; }
    pgas_ctx* c = new pgas_ctx();
    c->device = desc->device;
    DeviceGuard guard(desc->device);
    int dev_now = -1;
    int rc = (hipGetDevice(&dev_now) == hipSuccess && dev_now == desc->device) ? create_impl(desc, c) : PGAS_E_HIP;
    if (rc == PGAS_E_HIP && c->err.empty()) c->err = "pgas_create: cannot select the HIP device";
    if (rc != PGAS_OK) {
        g_create_error = c->err;
        pgas_destroy(c);
        *out = nullptr;
        return rc;
    }
    *out = c;
    return PGAS_OK;
}

void pgas_destroy(pgas_ctx* c) {
    if (!c) return;
    DeviceGuard guard(c->device);
    hipFree(c->d_y); hipFree(c->d_u); hipFree(c->d_idx); hipFree(c->d_pos); hipFree(c->d_qdesc); hipFree(c->d_mxi); hipFree(c->d_gimg); hipFree(c->d_m0L0); hipFree(c->d_ref);
    if (c->graph_exec) (void)hipGraphExecDestroy(c->graph_exec);
    if (c->graph) (void)hipGraphDestroy(c->graph);
    if (c->ev_g0) (void)hipEventDestroy(c->ev_g0);
    if (c->ev_g1) (void)hipEventDestroy(c->ev_g1);
    if (c->sG) (void)hipStreamDestroy(c->sG);
    hipFree(c->d_znoise); hipFree(c->d_duo);
    hipFree(c->d_tp); hipFree(c->d_sp); hipFree(c->d_ures); hipFree(c->d_uanc); hipFree(c->d_refbuf); hipFree(c->d_trajbuf);
    hipFree(c->d_G); hipFree(c->logw_last); hipFree(c->logw_trace); hipFree(c->d_bt);
    for (RowStore& r : c->rs) r.release();
    hipFree(c->segk_g[0]); hipFree(c->segk_g[1]); hipFree(c->segs_g[0]); hipFree(c->segs_g[1]);
    hipFree(c->d_phi); hipFree(c->d_syrk_ws); hipFree(c->aux_buf); hipFree(c->d_fail); hipFree(c->ws_partial); hipFree(c->d_sync);
    if (c->comm && rccl().destroy) rccl().destroy(c->comm);
    for (hipEvent_t e : c->ev) hipEventDestroy(e);
    for (hipEvent_t e : c->evp) hipEventDestroy(e);
    for (hipEvent_t e : c->ev_chunk) hipEventDestroy(e);
    for (hipEvent_t e : c->ev_bdone) hipEventDestroy(e);
    if (c->ev_start) hipEventDestroy(c->ev_start);
    if (c->ev_done) hipEventDestroy(c->ev_done);
    if (c->sB) hipStreamDestroy(c->sB);
    free_scanbufs(&c->sb[0]); free_scanbufs(&c->sb[1]);
    (void)hipGetLastError();  // do not leave a sticky error behind for the next context
    delete c;
}

static int pack_params(pgas_ctx* c, const double* A_dev, const double* S_dev, hipStream_t st) {
    c->tp.G = c->d_G;
    HIPCHK(c, hipMemsetAsync(c->d_G, 0, c->gtotal * sizeof(double), st));
    const int64_t n = (int64_t)c->md.M * c->md.nx;
    hipLaunchKernelGGL(k_pack, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, A_dev, c->d_pos, c->md.M, c->md.nx, c->md.nrm, c->d_G, c->gtotal, c->tp, S_dev, c->d_tp);
    KCHK(c, "k_pack");
    if (c->d_mxi) {
        hipLaunchKernelGGL(k_pack_mx, dim3((unsigned)((c->mx_slots * 64 + 255) / 256)), dim3(256), 0, st, (const double*)c->d_G, (const MxInfo*)c->d_mxi, c->d_gimg);
        KCHK(c, "k_pack_mx");
    }
    c->have_params = true;
    return PGAS_OK;
}

int pgas_set_params(pgas_ctx* c, const double* A_dev, const double* LS_host, const double* LSinv_host, double cS, void* stream) {
    if (!c) return PGAS_E_ARG;
    if (!A_dev || !LS_host || !LSinv_host) FAIL(c, PGAS_E_ARG, "pgas_set_params: NULL argument");
    DeviceGuard guard(c->device);
    const int nx = c->md.nx;
    for (int i = 0; i < 4; ++i) { c->tp.LS[i] = 0.0; c->tp.LSinv[i] = 0.0; }
    for (int k = 0; k < nx; ++k)
        for (int l = 0; l <= k; ++l) {
            c->tp.LS[k * nx + l] = LS_host[k * nx + l];
            c->tp.LSinv[k * nx + l] = LSinv_host[k * nx + l];
        }
    c->tp.cS = cS;
    return pack_params(c, A_dev, nullptr, (hipStream_t)stream);
}

/* The same with error_cov on the DEVICE (S_dev (nx,nx) row-major, symmetric positive definite): its Cholesky factor, the factor's
 * inverse and the normalising constant are formed by the pack kernel itself, so a Gibbs iteration (sample_params -> sweep) involves
 * no host round trip.  pgas_get_params reads back what the kernels will use (synchronises; tests). */
int pgas_set_params_dev(pgas_ctx* c, const double* A_dev, const double* S_dev, void* stream) {
    if (!c) return PGAS_E_ARG;
    if (!A_dev || !S_dev) FAIL(c, PGAS_E_ARG, "pgas_set_params_dev: NULL argument");
    DeviceGuard guard(c->device);
    return pack_params(c, A_dev, S_dev, (hipStream_t)stream);
}

int pgas_get_params(pgas_ctx* c, double* LS_host, double* LSinv_host, double* cS, void* stream) {
    if (!c) return PGAS_E_ARG;
    if (!LS_host || !LSinv_host || !cS) FAIL(c, PGAS_E_ARG, "pgas_get_params: NULL argument");
    if (!c->have_params) FAIL(c, PGAS_E_STATE, "pgas_get_params: call pgas_set_params first");
    DeviceGuard guard(c->device);
    HIPCHK(c, hipStreamSynchronize((hipStream_t)stream));
    TransParams tp;
    HIPCHK(c, hipMemcpy(&tp, c->d_tp, sizeof tp, hipMemcpyDeviceToHost));
    const int nx = c->md.nx;
    for (int k = 0; k < nx * nx; ++k) { LS_host[k] = tp.LS[k]; LSinv_host[k] = tp.LSinv[k]; }
    *cS = tp.cS;
    return PGAS_OK;
}

int pgas_basis_eval(pgas_ctx* c, const double* x_dev, int64_t np, int32_t t, double* phi_dev, void* stream) {
    if (!c) return PGAS_E_ARG;
    if (!x_dev || !phi_dev || np < 0 || t < 0 || t >= c->md.T) FAIL(c, PGAS_E_ARG, "pgas_basis_eval: bad argument");
    if (np == 0) return PGAS_OK;
    DeviceGuard guard(c->device);
    hipLaunchKernelGGL(c->basis, dim3((unsigned)((np + PG_BLK - 1) / PG_BLK)), dim3(PG_BLK), 0, (hipStream_t)stream, c->md, c->d_idx, x_dev, np, t, phi_dev);
    KCHK(c, "k_basis_eval");
    return PGAS_OK;
}

int pgas_aux_states(pgas_ctx* c, const double* x_dev, int32_t t, double* aux_dev, void* stream) {
    if (!c) return PGAS_E_ARG;
    if (!x_dev || !aux_dev || t < 0 || t >= c->md.T) FAIL(c, PGAS_E_ARG, "pgas_aux_states: bad argument");
    if (!c->have_params) FAIL(c, PGAS_E_STATE, "pgas_aux_states: call pgas_set_params first");
    DeviceGuard guard(c->device);
    hipLaunchKernelGGL(c->var.aux, dim3(c->md.nseg), dim3(PG_BLK), 0, (hipStream_t)stream, c->md, (const TransParams*)c->d_tp, t, x_dev, aux_dev);
    KCHK(c, "k_aux");
    return PGAS_OK;
}

int pgas_init_state(pgas_ctx* c, uint64_t seed, const double* ref0_host, double* x0_dev, void* stream) {
    if (!c) return PGAS_E_ARG;
    if (!ref0_host || !x0_dev) FAIL(c, PGAS_E_ARG, "pgas_init_state: NULL argument");
    DeviceGuard guard(c->device);
    hipStream_t st = (hipStream_t)stream;
    HIPCHK(c, hipMemcpyAsync(c->d_ref, ref0_host, c->md.nx * sizeof(double), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(c->init, dim3((c->md.N + PG_BLK - 1) / PG_BLK), dim3(PG_BLK), 0, st, c->md, seed, (const SweepParams*)nullptr, c->d_m0L0, c->d_ref, x0_dev);
    KCHK(c, "k_init");
    return PGAS_OK;
}

// ---- launch helpers of the weight recursion (pgas_resample.hip.h) ---------------------------------------------------
static Peers peers_for(const pgas_ctx* c, int parity) {
    Peers p = c->peers;
    for (int r = 0; r < p.world; ++r) {
        p.c1[r] = c->peer_c1[parity][r];
        p.c2[r] = c->peer_c2[parity][r];
    }
    return p;
}

// this context's own buffers as "rank" c->rank of the peer table (single device: rank 0)
static void install_own_peers(pgas_ctx* c) {
    const int r = c->rank;
    for (int k = 0; k < PG_RB_NKIND; ++k) c->peer_blk[r][k].assign(c->rs[k].blk.begin(), c->rs[k].blk.end());
    c->bt_dirty = true;
    for (int i = 0; i < 2; ++i) {
        c->peer_c1[i][r] = c->sb[i].c1;
        c->peer_c2[i][r] = c->sb[i].c2;
    }
}

// row t of trace `kind` on rank r, as addressable from this device (same block layout on every rank: equal N_local and T)
static const char* peer_row(const pgas_ctx* c, int r, int kind, int t) {
    const RowStore& s = c->rs[kind];
    return c->peer_blk[r][kind][(size_t)t >> s.shift] + (size_t)(t & ((1 << s.shift) - 1)) * s.row_bytes;
}

// device copy of the (rank, {x, anc}, block) table the ancestor chase walks
static int upload_bt_table(pgas_ctx* c, hipStream_t st, BtTab* out) {
    const int nb = std::max(c->rs[PG_RB_X].nblk, c->rs[PG_RB_ANC].nblk);
    const size_t entries = (size_t)c->world * 2 * nb;
    if (c->bt_dirty || c->d_bt_entries != entries) {
        std::vector<const void*> h(entries, nullptr);
        for (int r = 0; r < c->world; ++r) {
            const auto& bx = c->peer_blk[r][PG_RB_X];
            const auto& ba = c->peer_blk[r][PG_RB_ANC];
            for (size_t b = 0; b < bx.size(); ++b) h[((size_t)r * 2) * nb + b] = bx[b];
            for (size_t b = 0; b < ba.size(); ++b) h[((size_t)r * 2 + 1) * nb + b] = ba[b];
        }
        HIPCHK(c, hipStreamSynchronize(st));   // an earlier chase may still be reading the old table
        if (c->d_bt_entries != entries) {
            hipFree(c->d_bt);
            c->d_bt = nullptr; c->d_bt_entries = 0;
            HIPCHK(c, hipMalloc(&c->d_bt, entries * sizeof(void*)));
            c->d_bt_entries = entries;
        }
        HIPCHK(c, hipMemcpy(c->d_bt, h.data(), entries * sizeof(void*), hipMemcpyHostToDevice));
        c->bt_dirty = false;
    }
    out->blk = c->d_bt;
    out->nblk_max = nb;
    out->shift_x = c->rs[PG_RB_X].shift;
    out->shift_anc = c->rs[PG_RB_ANC].shift;
    out->entries = (int32_t)entries;
    return PGAS_OK;
}

static int launch_backtrace(pgas_ctx* c, const UpperHdr* hdr, double* traj_dev, hipStream_t st) {
    BtTab tab;
    int rc = upload_bt_table(c, st, &tab);
    if (rc) return rc;
    const size_t lds = tab.entries <= 2048 ? (size_t)tab.entries * sizeof(void*) : 0;
    hipLaunchKernelGGL(k_backtrace, dim3(1), dim3(64), lds, st, c->md.N, c->md.T, c->md.nx, tab, c->world, hdr, traj_dev);
    KCHK(c, "k_backtrace");
    return PGAS_OK;
}

// group records of `ncdf` CDFs from the (gathered) segment partials of sb
static int launch_groups(pgas_ctx* c, const ScanBufs& sb, int ncdf, hipStream_t st) {
    const int n1 = (c->md.nseg_g + PG_GRP - 1) / PG_GRP;
    hipLaunchKernelGGL(k_groups, dim3((n1 * ncdf + 3) / 4), dim3(256), 0, st, c->md.nseg_g, ncdf, sb);
    KCHK(c, "k_groups");
    return PGAS_OK;
}

// what = 0: ancestor of the conditioned particle (step API), 1: final index
static int launch_count(pgas_ctx* c, const ScanBufs& sb, int parity, int what, double u, hipStream_t st) {
    // what = 1 (final index of a sweep): the uniform is the running sweep's, read from the device (u is ignored)
    hipLaunchKernelGGL(k_count, dim3(1), dim3(PG_BLK), 0, st, c->md.Ng, c->md.nseg_g, sb, peers_for(c, parity), what, u, what == 1 ? (const SweepParams*)c->d_sp : (const SweepParams*)nullptr);
    KCHK(c, "k_count");
    return PGAS_OK;
}

// k_step<LOCAL> (every workgroup scans all groups itself, no k_groups launch between the steps) is an option, not the default:
// measured at N = 2^20 the k_groups path is 9 % faster (81.4 against 90.2 ms per sweep; the redundant scans cost more vector
// issue than the extra tiny launch costs latency).  PGAS_OPT_LOCAL_GROUPS selects it.
static bool sweep_is_local(const pgas_ctx* c) { return c->local_groups && c->world == 1 && !c->sharded && c->md.nseg_g <= PG_LOCAL_NSEG && !c->force_slow; }

// PGAS_OPT_TAIL_GROUPS (single device only; sharded sweeps need the all-gather first): the group scans ride in k_step's tail, done by
// the workgroup that completes a group, instead of a k_groups launch between the steps.  Correct and bit-identical, but measured
// SLOWER (88.5 against 84.9 ms per sweep at N = 2^20: the write-through stores, the drain and the returning atomic at the end of
// every workgroup cost more than the ~4 us launch they replace), so it is off by default and kept as an experiment.
static bool sweep_tail_groups(const pgas_ctx* c) { return !c->sharded && c->world == 1 && c->tail_groups; }

// launch t in [1, T] of the sweep: resamples step t-1 (t > 1), scans step t (t < T)
static int launch_step(pgas_ctx* c, int t, bool local, hipStream_t st, hipEvent_t e0, hipEvent_t e1) {
    const DevModel& md = c->md;
    const int N = md.N, T = md.T;
    StepArgs ar;
    ar.t = t;
    ar.mode = (t < T ? PG_RS_SCAN : 0) | (t > 1 ? PG_RS_SEARCH : 0);
    ar.sp = c->d_sp;
    ar.u_res = c->d_ures;
    ar.u_anc = c->d_uanc;
    ar.la_t = t < T ? (const double*)c->rs[PG_RB_LA].row(t) : (const double*)nullptr;
    ar.h_t = t < T ? (const double*)c->rs[PG_RB_H].row(t) : (const double*)nullptr;
    ar.ln_prev = (const double*)c->rs[PG_RB_LN].row(t - 1);
    ar.anc_in = AncIn{};
    ar.anc_in.has_prev = t > 2 ? 1 : 0;
    if (t > 1)
        for (int r = 0; r < c->world; ++r) {   // rows of step s = t-1 (and s-1) on every rank
            ar.anc_in.la_s[r] = (const double*)peer_row(c, r, PG_RB_LA, t - 1);
            ar.anc_in.h_s[r] = (const double*)peer_row(c, r, PG_RB_H, t - 1);
            if (t > 2) {
                ar.anc_in.ln_p[r] = (const double*)peer_row(c, r, PG_RB_LN, t - 2);
                ar.anc_in.la_p[r] = (const double*)peer_row(c, r, PG_RB_LA, t - 2);
                ar.anc_in.anc_p[r] = (const int32_t*)peer_row(c, r, PG_RB_ANC, t - 3);
            }
        }
    ar.anc_out = t > 1 ? (int32_t*)c->rs[PG_RB_ANC].row(t - 2) : (int32_t*)nullptr;
    ar.logw_out = t == T ? c->logw_last : ((c->logw_trace && t > 1) ? c->logw_trace + (size_t)(t - 1) * N : (double*)nullptr);
    const ScanBufs& sp = c->sb[(t - 1) & 1];
    const ScanBufs& sn = c->sb[t & 1];
    const Peers pr = peers_for(c, (t - 1) & 1);
    auto kern = local ? k_step<PG_WM_LOCAL, false> : (sweep_tail_groups(c) ? k_step<PG_WM_GROUPS, true> : k_step<PG_WM_GROUPS, false>);
    // dispatch-attached events only when this launch is timed: the plain launch is what a stream capture records
    if (e0 || e1) hipExtLaunchKernelGGL(kern, dim3(md.nseg + 1), dim3(PG_BLK), 0, st, e0, e1, 0, md, ar, sp, sn, pr);
    else hipLaunchKernelGGL(kern, dim3(md.nseg + 1), dim3(PG_BLK), 0, st, md, ar, sp, sn, pr);
    KCHK(c, "k_step");
    return PGAS_OK;
}

int pgas_step(pgas_ctx* c, int32_t t, uint64_t seed, const double* logw_dev, const double* x_dev, const double* ref_t_host,
              double* logw_new_dev, double* x_new_dev, int32_t* anc_dev, void* stream) {
    if (!c) return PGAS_E_ARG;
    if (!x_dev || !ref_t_host || !logw_new_dev || !x_new_dev || !anc_dev) FAIL(c, PGAS_E_ARG, "pgas_step: NULL argument");
    if (t < 0 || t >= c->md.T) FAIL(c, PGAS_E_ARG, "pgas_step: t = %d outside [0, %d)", t, c->md.T);
    if (!c->have_params) FAIL(c, PGAS_E_STATE, "pgas_step: call pgas_set_params first");
    if (c->sharded) FAIL(c, PGAS_E_STATE, "pgas_step: not available on a shard context");
    DeviceGuard guard(c->device);
    hipStream_t st = (hipStream_t)stream;
    const DevModel& md = c->md;
    install_own_peers(c);
    HIPCHK(c, hipMemcpyAsync(c->d_ref, ref_t_host, md.nx * sizeof(double), hipMemcpyHostToDevice, st));
    if (c->corrected && !c->aux_buf) HIPCHK(c, hipMalloc(&c->aux_buf, (size_t)md.N * md.nx * sizeof(double)));
    hipLaunchKernelGGL(c->var.front, dim3(md.nseg), dim3(PG_BLK), 0, st, md, (const TransParams*)c->d_tp, t, seed, x_dev, logw_dev, c->d_ref, c->corrected,
                       c->corrected ? c->aux_buf : x_new_dev, c->sb[0]);
    KCHK(c, "k_front");
    int rc = launch_groups(c, c->sb[0], 2, st);
    if (rc) return rc;
    rc = launch_count(c, c->sb[0], 0, 0, pgas_rng_uniform(seed, PGAS_STREAM_ANCESTOR, (uint32_t)t), st);
    if (rc) return rc;
    const Peers pr = peers_for(c, 0);
    if (c->corrected) {
        hipLaunchKernelGGL(c->back_corrected, dim3(md.nseg), dim3(PG_BLK), 0, st, md, (const TransParams*)c->d_tp, t, seed,
                           pgas_rng_uniform(seed, PGAS_STREAM_RESAMPLE, (uint32_t)t), c->aux_buf, c->d_ref, c->sb[0], pr, anc_dev, x_new_dev, logw_new_dev);
    } else {
        hipLaunchKernelGGL(c->back, dim3(md.nseg), dim3(PG_BLK), 0, st, md, t,
                           pgas_rng_uniform(seed, PGAS_STREAM_RESAMPLE, (uint32_t)t), x_new_dev, c->sb[0], pr, anc_dev, logw_new_dev);
    }
    KCHK(c, "k_back");
    return PGAS_OK;
}

// rows x row_bytes as one allocation (block_bytes == 0) or as power-of-two runs of rows of at most block_bytes each
static int alloc_rows(pgas_ctx* c, RowStore* s, int rows, size_t row_bytes, size_t block_bytes) {
    s->release();
    s->rows = rows; s->row_bytes = row_bytes;
    if (block_bytes == 0) {
        s->contiguous = true; s->shift = 30; s->nblk = 1;
        void* p = nullptr;
        HIPCHK(c, hipMalloc(&p, (size_t)rows * row_bytes));
        s->own.push_back(p);
        s->blk.push_back((char*)p);
        return PGAS_OK;
    }
    if (row_bytes > block_bytes) FAIL(c, PGAS_E_ARG, "trace rows of %zu bytes do not fit blocks of %zu bytes", row_bytes, block_bytes);
    int shift = 0;
    while (shift < 30 && ((size_t)2 << shift) * row_bytes <= block_bytes) ++shift;
    s->contiguous = false; s->shift = shift; s->nblk = (rows + (1 << shift) - 1) >> shift;
    for (int b = 0; b < s->nblk; ++b) {
        void* p = nullptr;
        HIPCHK(c, hipMalloc(&p, (size_t)s->rows_in_block(b) * row_bytes));
        s->own.push_back(p);
        s->blk.push_back((char*)p);
    }
    return PGAS_OK;
}

static int ensure_traces(pgas_ctx* c, bool blocked) {
    if (c->have_traces) return PGAS_OK;
    const DevModel& md = c->md;
    const size_t np = (size_t)md.nseg * PGAS_SEG;
    const size_t bb = c->trace_block_bytes ? c->trace_block_bytes : (blocked ? PG_TRACE_BLOCK_BYTES : 0);
    int rc = alloc_rows(c, &c->rs[PG_RB_X], md.T, (size_t)md.N * md.nx * sizeof(double), bb);
    if (!rc) rc = alloc_rows(c, &c->rs[PG_RB_ANC], md.T > 1 ? md.T - 1 : 1, (size_t)md.N * sizeof(int32_t), bb);
    for (int k : {PG_RB_LA, PG_RB_H, PG_RB_LN})
        if (!rc) rc = alloc_rows(c, &c->rs[k], md.T, np * sizeof(double), bb);
    if (rc) {
        for (RowStore& r : c->rs) r.release();
        return rc;
    }
    HIPCHK(c, hipMalloc(&c->logw_last, (size_t)md.N * sizeof(double)));
    if (c->keep_logw) {
        HIPCHK(c, hipMalloc(&c->logw_trace, (size_t)md.T * md.N * sizeof(double)));
        HIPCHK(c, hipMemset(c->logw_trace, 0, (size_t)md.T * md.N * sizeof(double)));
    }
    c->have_traces = true;
    install_own_peers(c);
    return PGAS_OK;
}

// k_propagate for time steps [t0, t1), optionally carrying dispatch-attached events
static int launch_propagate(pgas_ctx* c, int t0, int t1, const double* ref_dev, hipStream_t st, bool timed) {
    const int spw = c->var.PPT / PG_PPT;   // segments per k_propagate workgroup
    const dim3 grid((c->md.nseg + spw - 1) / spw), blk(PG_BLK);
    size_t lds = c->overlap ? c->prop_lds : 0;
    const bool mx = c->d_mxi && c->use_mx;
    if (mx) {   // operand image + one transposed sine table per wave
        lds = (size_t)c->mx_slots * 64 * sizeof(double) + (size_t)(PG_BLK / 64) * 64 * PG_MX_TSTRIDE * sizeof(double);
    } else if (c->md.D == 3) {   // the LDS copy of the coefficient tensor (k_propagate, 3-D variants)
        const size_t need = (size_t)c->gtotal * sizeof(double);
        if (need > 64 * 1024) FAIL(c, PGAS_E_ARG, "k_propagate: coefficient tensor of %zu bytes does not fit the LDS budget", need);
        lds = lds > need ? lds : need;
    }
    // a launch writes rows [ta, tb) that lie in ONE block of every trace (blocks are power-of-two runs of rows, the hand-off rows'
    // at least as long as the state rows'): a chunk that straddles a block boundary is split there
    const int sh = std::min(c->rs[PG_RB_X].shift, c->rs[PG_RB_LA].shift);
    for (int ta = t0; ta < t1;) {
        const int tb = std::min(t1, ((ta >> sh) + 1) << sh);
        const prop_fn prop = (tb == ta + 1 && c->var.prop_one) ? c->var.prop_one : c->var.prop;
        const propmx_fn pmx = tb == ta + 1 ? k_propagate_mx<2, 12, PG_WMX, PG_PPT3, true, PG_MX_DESC_BALL729> : k_propagate_mx<2, 12, PG_WMX, PG_PPT3, false, PG_MX_DESC_BALL729>;
        const double* x_prev = (const double*)c->rs[PG_RB_X].row(ta - 1);
        double* x_rows = (double*)c->rs[PG_RB_X].row(ta);
        double* la = (double*)c->rs[PG_RB_LA].row(ta);
        double* h = (double*)c->rs[PG_RB_H].row(ta);
        double* ln = (double*)c->rs[PG_RB_LN].row(ta);
        if (timed) {
            while ((int)c->evp.size() < c->evp_used + 2) {
                hipEvent_t e;
                HIPCHK(c, hipEventCreate(&e));
                c->evp.push_back(e);
            }
            // start/stop events bound to the dispatch itself: they carry the kernel's own begin/end timestamps
            if (mx)
                hipExtLaunchKernelGGL(pmx, grid, blk, lds, st, c->evp[c->evp_used], c->evp[c->evp_used + 1], 0, c->md, (const TransParams*)c->d_tp,
                                      (const double*)c->d_G, (const SweepParams*)c->d_sp, ta, tb, x_prev, x_rows, ref_dev, la, h, ln, (const MxInfo*)c->d_mxi, (const double*)c->d_gimg);
            else
            hipExtLaunchKernelGGL(prop, grid, blk, lds, st, c->evp[c->evp_used], c->evp[c->evp_used + 1], 0, c->md, (const TransParams*)c->d_tp,
                                  (const double*)c->d_G, (const SweepParams*)c->d_sp, ta, tb, x_prev, x_rows, ref_dev, la, h, ln);
            c->evp_used += 2;
        } else if (mx) {
            hipLaunchKernelGGL(pmx, grid, blk, lds, st, c->md, (const TransParams*)c->d_tp, (const double*)c->d_G, (const SweepParams*)c->d_sp, ta, tb, x_prev, x_rows, ref_dev, la, h, ln,
                               (const MxInfo*)c->d_mxi, (const double*)c->d_gimg);
        } else {
            hipLaunchKernelGGL(prop, grid, blk, lds, st, c->md, (const TransParams*)c->d_tp, (const double*)c->d_G, (const SweepParams*)c->d_sp, ta, tb, x_prev, x_rows, ref_dev, la, h, ln);
        }
        KCHK(c, "k_propagate");
        ta = tb;
    }
    return PGAS_OK;
}

static int ensure_side_stream(pgas_ctx* c, int nchunk) {
    if (!c->sB) {
        int lo = 0, hi = 0;
        HIPCHK(c, hipDeviceGetStreamPriorityRange(&lo, &hi));
        // round 1 measured the chain stream at LOW priority as best; with the round-2 kernels the priority makes no difference
        const char* pe = getenv("PGAS_CHAIN_PRIO");   // development knob: 1 = highest, 2 = default priority
        const int pr = pe && pe[0] == '1' ? hi : (pe && pe[0] == '2' ? 0 : lo);
        HIPCHK(c, hipStreamCreateWithPriority(&c->sB, hipStreamNonBlocking, pr));
        HIPCHK(c, hipEventCreateWithFlags(&c->ev_start, hipEventDisableTiming));
        HIPCHK(c, hipEventCreateWithFlags(&c->ev_done, hipEventDisableTiming));
    }
    while ((int)c->ev_chunk.size() < nchunk) {
        hipEvent_t e;
        HIPCHK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        c->ev_chunk.push_back(e);
    }
    return PGAS_OK;
}

// One collective of the sharded sweep: the (2, nsegp) segment partials of every rank -> (world, 2, nsegp).  RCCL by default
// (both arrays in one group, on the sweep's stream); a host callback when the caller installed one (tests on one device).
static int shard_all_gather(pgas_ctx* c, int parity, hipStream_t st);

// seed, epoch and the T + 1 resampling / ancestor uniforms of a new sweep into device memory (what its kernels read them from)
static int sweep_begin(pgas_ctx* c, uint64_t seed, hipStream_t st) {
    ++c->epoch;
    hipLaunchKernelGGL(k_sweep_begin, dim3((unsigned)((c->md.T + 1 + 255) / 256)), dim3(256), 0, st, seed, c->epoch, c->md.T, c->d_sp, c->d_ures, c->d_uanc);
    KCHK(c, "k_sweep_begin");
    return PGAS_OK;
}

// The time loop shared by pgas_sweep and pgas_shard_sweep: pipeline A (k_propagate, caller's stream) runs ahead chunk by chunk,
// pipeline B (k_step [+ all-gather + k_groups]) follows on the internal stream, gated by one event per chunk.
static int run_time_loop(pgas_ctx* c, const double* ref_dev, int chunk, hipStream_t st) {
    const DevModel& md = c->md;
    const int T = md.T;
    const bool local = sweep_is_local(c);
    const int nchunk = (T - 1 + chunk - 1) / chunk;
    hipStream_t sB = st;
    if (c->overlap) {
        int rc = ensure_side_stream(c, nchunk);
        if (rc) return rc;
        sB = c->sB;
        HIPCHK(c, hipEventRecord(c->ev_start, st));
        HIPCHK(c, hipStreamWaitEvent(sB, c->ev_start, 0));
    }
    if (c->profiling && (int)c->ev.size() < 2 * T) {
        const size_t old = c->ev.size();
        c->ev.resize(2 * (size_t)T);
        for (size_t i = old; i < c->ev.size(); ++i) HIPCHK(c, hipEventCreate(&c->ev[i]));
    }
    // Launches are issued in groups of `ev_stride` chunks, pipeline A first, so both device queues stay fed.  One event per
    // group (not per chunk) gates pipeline B: a marker packet between two consecutive k_propagate launches costs pipeline A a
    // command-processor round trip per step, and B trails A by more than a group anyway.
    const int stride = c->ev_stride > 0 ? c->ev_stride : 1;
    // PGAS_OPT_MAX_LEAD: pipeline A may run at most `max_lead` groups ahead of pipeline B, so that the hand-off rows B reads are
    // still in the memory-side cache when it gets to them (0 = unbounded)
    const int lead = sB != st ? c->max_lead : 0;
    const int ngroups = (nchunk + stride - 1) / stride;
    if (lead > 0)
        while ((int)c->ev_bdone.size() < ngroups) {
            hipEvent_t e;
            HIPCHK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
            c->ev_bdone.push_back(e);
        }
    for (int c0 = 0; c0 < nchunk; c0 += stride) {
        const int c1 = c0 + stride < nchunk ? c0 + stride : nchunk;
        const int gi = c0 / stride;
        if (lead > 0 && gi >= lead) HIPCHK(c, hipStreamWaitEvent(st, c->ev_bdone[gi - lead], 0));
        for (int ci = c0; ci < c1; ++ci) {
            const int t0 = 1 + ci * chunk, t1 = t0 + chunk < T ? t0 + chunk : T;
            int rc = launch_propagate(c, t0, t1, ref_dev, st, c->profiling && (ci % c->prof_stride) == 0);
            if (rc) return rc;
        }
        if (sB != st) {
            HIPCHK(c, hipEventRecord(c->ev_chunk[c0], st));
            HIPCHK(c, hipStreamWaitEvent(sB, c->ev_chunk[c0], 0));
        }
        // launch t resamples step t-1 (t > 1) and scans step t (t < T); the last group also runs launch T
        const int tbeg = 1 + c0 * chunk;
        const int tend = c1 == nchunk ? T + 1 : 1 + c1 * chunk;
        for (int t = tbeg; t < tend; ++t) {
            const bool timed = c->profiling && t < T && (t % c->prof_stride) == 0;
            hipEvent_t e0 = timed ? c->ev[c->ev_used] : (hipEvent_t) nullptr, e1 = timed ? c->ev[c->ev_used + 1] : (hipEvent_t) nullptr;
            if (timed) c->ev_used += 2;
            int rc = launch_step(c, t, local, sB, e0, e1);
            if (rc) return rc;
            if (!local && t < T && !sweep_tail_groups(c)) {
                if (c->sharded) {
                    rc = shard_all_gather(c, t & 1, sB);   // the one collective of the step, stream-ordered: no host round trip
                    if (rc) return rc;
                }
                rc = launch_groups(c, c->sb[t & 1], 2, sB);
                if (rc) return rc;
            }
        }
        if (lead > 0) HIPCHK(c, hipEventRecord(c->ev_bdone[gi], sB));
    }
    if (sB != st) {
        HIPCHK(c, hipEventRecord(c->ev_done, sB));
        HIPCHK(c, hipStreamWaitEvent(st, c->ev_done, 0));
    }
    return PGAS_OK;
}

// final index (src/PGAS.py:224-225) and back-trace (src/Filtering.py:40-55)
static int run_final(pgas_ctx* c, double* traj_dev, hipStream_t st) {
    const DevModel& md = c->md;
    const int T = md.T;
    const ScanBufs& sf = c->sb[T & 1];
    hipLaunchKernelGGL(k_segscan, dim3(md.nseg), dim3(PG_BLK), 0, st, md.N, c->logw_last, sf);
    KCHK(c, "k_segscan");
    int rc;
    if (c->sharded) {
        rc = shard_all_gather(c, T & 1, st);
        if (rc) return rc;
    }
    rc = launch_groups(c, sf, 1, st);
    if (rc) return rc;
    rc = launch_count(c, sf, T & 1, 1, 0.0, st);
    if (rc) return rc;
    return launch_backtrace(c, sf.hdr, traj_dev, st);
}

// everything of a default-mode sweep after sweep_begin: x_0, the time loop, the final index and the back-trace
static int sweep_body(pgas_ctx* c, const double* ref_dev, double* traj_dev, int chunk, hipStream_t st) {
    const DevModel& md = c->md;
    hipLaunchKernelGGL(c->init, dim3((md.N + PG_BLK - 1) / PG_BLK), dim3(PG_BLK), 0, st, md, (uint64_t)0, (const SweepParams*)c->d_sp, c->d_m0L0, ref_dev,
                       (double*)c->rs[PG_RB_X].row(0));
    KCHK(c, "k_init");
    if (md.T == 1) {
        HIPCHK(c, hipMemsetAsync(c->logw_last, 0, md.N * sizeof(double), st));
    } else {
        int rc = run_time_loop(c, ref_dev, chunk, st);
        if (rc) return rc;
        if (c->logw_trace)
            HIPCHK(c, hipMemcpyAsync(c->logw_trace + (size_t)(md.T - 1) * md.N, c->logw_last, md.N * sizeof(double), hipMemcpyDeviceToDevice, st));
    }
    return run_final(c, traj_dev, st);
}

// Capture sweep_body once (reference trajectory and result in the library's own buffers) and keep the instantiated graph; any
// failure leaves the context on the eager path for good.  Returns true when c->graph_exec is ready for `key`.
static bool sweep_graph_ready(pgas_ctx* c, const int (&key)[6], int chunk) {
    if (c->graph_failed) return false;
    bool same = c->graph_exec != nullptr;
    for (int i = 0; i < 6; ++i) same = same && c->graph_key[i] == key[i];
    if (same) return true;
    auto give_up = [&](const char* what, hipError_t e) {
        const char* dbg = getenv("PGAS_GRAPH_DEBUG");
        if (dbg && dbg[0] == '1') fprintf(stderr, "pgas: sweep graph disabled: %s: %s\n", what, hipGetErrorString(e));
        (void)hipGetLastError();
        c->graph_failed = 1;
        return false;
    };
    if (c->graph_exec) { (void)hipGraphExecDestroy(c->graph_exec); c->graph_exec = nullptr; }
    if (c->graph) { (void)hipGraphDestroy(c->graph); c->graph = nullptr; }
    hipError_t e = hipSuccess;
    if (!c->sG) {
        e = hipStreamCreateWithFlags(&c->sG, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_g0, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_g1, hipEventDisableTiming);
        if (e != hipSuccess) return give_up("stream / event creation", e);
    }
    // nothing that allocates or synchronises may run inside the capture: side stream, events and the back-trace table first
    const int nchunk = c->md.T > 1 ? (c->md.T - 1 + chunk - 1) / chunk : 1;
    BtTab tab;
    if ((c->overlap && ensure_side_stream(c, nchunk)) || upload_bt_table(c, c->sG, &tab)) return give_up("set-up before the capture", hipErrorUnknown);
    e = hipStreamBeginCapture(c->sG, hipStreamCaptureModeThreadLocal);
    if (e != hipSuccess) return give_up("hipStreamBeginCapture", e);
    const int rc = sweep_body(c, c->d_refbuf, c->d_trajbuf, chunk, c->sG);
    hipGraph_t g = nullptr;
    e = hipStreamEndCapture(c->sG, &g);
    if (rc || e != hipSuccess || !g) {
        if (g) (void)hipGraphDestroy(g);
        return give_up(rc ? c->err.c_str() : "hipStreamEndCapture", e);
    }
    hipGraphExec_t ex = nullptr;
    e = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
    if (e != hipSuccess || !ex) {
        (void)hipGraphDestroy(g);
        return give_up("hipGraphInstantiate", e);
    }
    c->graph = g;
    c->graph_exec = ex;
    for (int i = 0; i < 6; ++i) c->graph_key[i] = key[i];
    return true;
}

int pgas_sweep(pgas_ctx* c, uint64_t seed, const double* ref_dev, double* traj_dev, void* stream) {
    if (!c) return PGAS_E_ARG;
    if (!ref_dev || !traj_dev) FAIL(c, PGAS_E_ARG, "pgas_sweep: NULL argument");
    if (!c->have_params) FAIL(c, PGAS_E_STATE, "pgas_sweep: call pgas_set_params first");
    if (c->sharded) FAIL(c, PGAS_E_STATE, "pgas_sweep: this context is a shard; drive it with pgas_shard_sweep");
    DeviceGuard guard(c->device);
    int rc = ensure_traces(c, false);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    const DevModel& md = c->md;
    const int N = md.N, T = md.T, nx = md.nx;
    const size_t row = (size_t)N * nx;
    const dim3 grid(md.nseg), blk(PG_BLK);
    auto xrow = [&](int t) { return (double*)c->rs[PG_RB_X].row(t); };
    c->ev_used = 0;
    c->evp_used = 0;
    if (c->corrected && T > 1) {
        // corrected mode: x_t depends on the ancestors a_t, so the step is a serial chain on one stream
        // (transition means + scans, group scans, reference ancestor, search + propagate + weights)
        rc = sweep_begin(c, seed, st);   // final-index uniform
        if (rc) return rc;
        hipLaunchKernelGGL(c->init, dim3((N + PG_BLK - 1) / PG_BLK), blk, 0, st, md, seed, (const SweepParams*)nullptr, c->d_m0L0, ref_dev, xrow(0));
        KCHK(c, "k_init");
        if (!c->aux_buf) HIPCHK(c, hipMalloc(&c->aux_buf, row * sizeof(double)));
        const Peers pr = peers_for(c, 0);
        for (int t = 1; t < T; ++t) {
            const double* lw_prev = t == 1 ? (const double*)nullptr : (c->logw_trace ? c->logw_trace + (size_t)(t - 1) * N : c->logw_last);
            double* lw_out = c->logw_trace ? c->logw_trace + (size_t)t * N : c->logw_last;
            hipLaunchKernelGGL(c->var.front, grid, blk, 0, st, md, (const TransParams*)c->d_tp, t, seed, xrow(t - 1), lw_prev, ref_dev + (size_t)t * nx, 1,
                               c->aux_buf, c->sb[0]);
            KCHK(c, "k_front");
            rc = launch_groups(c, c->sb[0], 2, st);
            if (rc) return rc;
            rc = launch_count(c, c->sb[0], 0, 0, pgas_rng_uniform(seed, PGAS_STREAM_ANCESTOR, (uint32_t)t), st);
            if (rc) return rc;
            hipLaunchKernelGGL(c->back_corrected, grid, blk, 0, st, md, (const TransParams*)c->d_tp, t, seed, pgas_rng_uniform(seed, PGAS_STREAM_RESAMPLE, (uint32_t)t),
                               c->aux_buf, ref_dev + (size_t)t * nx, c->sb[0], pr, (int32_t*)c->rs[PG_RB_ANC].row(t - 1), xrow(t), lw_out);
            KCHK(c, "k_back_corrected");
        }
        if (c->logw_trace)
            HIPCHK(c, hipMemcpyAsync(c->logw_last, c->logw_trace + (size_t)(T - 1) * N, N * sizeof(double), hipMemcpyDeviceToDevice, st));
        return run_final(c, traj_dev, st);
    }
    c->last_small = 0;
    c->last_graph = 0;
    if (c->use_small && N <= PGAS_SEG && !c->profiling && c->rs[PG_RB_X].contiguous && c->rs[PG_RB_ANC].contiguous) {
        // at most one segment of particles: the whole sweep -- x_0, T-1 steps, final index, back-trace -- is ONE launch of ONE workgroup
        rc = sweep_begin(c, seed, st);
        if (rc) return rc;
        if (!c->d_znoise) HIPCHK(c, hipMalloc(&c->d_znoise, (size_t)T * N * 2 * sizeof(double)));
        if (T > 1) {
            const int64_t nz = (int64_t)N * (T - 1);
            hipLaunchKernelGGL(k_small_noise, dim3((unsigned)((nz + 255) / 256)), dim3(256), 0, st, (const SweepParams*)c->d_sp, md.p0, N, T, c->d_znoise);
            KCHK(c, "k_small_noise");
        }
        const size_t lds = (size_t)c->gtotal * sizeof(double);   // the coefficient tensor, beside 41 KB of static LDS
        if (lds > 64 * 1024) FAIL(c, PGAS_E_ARG, "k_sweep_small: coefficient tensor of %zu bytes does not fit the LDS budget", lds);
        const int nri = N <= PG_BLK ? 0 : (N <= 2 * PG_BLK ? 1 : 2);
        if (c->use_small == 1) {   // default: two workgroups (PGAS_OPT_SMALL_SWEEP = 2: one)
            if (!c->d_duo) HIPCHK(c, hipMalloc(&c->d_duo, sizeof(DuoShared)));
            HIPCHK(c, hipMemsetAsync(c->d_duo, 0, 64, st));   // the two counters
            hipLaunchKernelGGL(c->var.duo[nri], dim3(2), dim3(PG_BLK), lds, st, md, (const TransParams*)c->d_tp, (const double*)c->d_G, (const SweepParams*)c->d_sp,
                               (const double*)c->d_ures, (const double*)c->d_uanc, (const double*)c->d_m0L0, ref_dev, (double*)c->rs[PG_RB_X].blk[0],
                               (int32_t*)c->rs[PG_RB_ANC].blk[0], c->logw_last, c->logw_trace, c->sb[T & 1].hdr, traj_dev, (const double*)c->d_znoise, c->d_duo);
            KCHK(c, "k_sweep_duo");
            c->last_small = 1;
            c->last_chunk = 1;
            return PGAS_OK;
        }
        hipLaunchKernelGGL(c->var.small[nri], dim3(1), dim3(PG_BLK), lds, st, md, (const TransParams*)c->d_tp, (const double*)c->d_G, (const SweepParams*)c->d_sp,
                           (const double*)c->d_ures, (const double*)c->d_uanc, (const double*)c->d_m0L0, ref_dev, (double*)c->rs[PG_RB_X].blk[0],
                           (int32_t*)c->rs[PG_RB_ANC].blk[0], c->logw_last, c->logw_trace, c->sb[T & 1].hdr, traj_dev, (const double*)c->d_znoise);
        KCHK(c, "k_sweep_small");
        c->last_small = 1;
        c->last_chunk = 1;
        return PGAS_OK;
    }
    // default chunk, measured (tools/ab_bench.py --chunk, tools/config_times.py): one step per k_propagate launch
    const int chunk = c->prop_chunk > 0 ? c->prop_chunk : (c->overlap ? 1 : T);
    c->last_chunk = chunk;
    // PGAS_OPT_GRAPH (default on): the ~6000 launches of a sweep are captured once and replayed; launches that carry timing events
    // (pgas_set_profiling) and the development knobs that wait across streams stay on the eager path
    const int key[6] = {chunk, c->overlap, c->ev_stride, sweep_is_local(c) ? 1 : (sweep_tail_groups(c) ? 2 : 0), c->prop_lds, c->keep_logw};
    const bool want_graph = c->use_graph == 1 || (c->use_graph < 0 && md.N <= PG_GRAPH_AUTO_N);
    if (want_graph && !c->profiling && c->max_lead == 0 && sweep_graph_ready(c, key, chunk)) {
        // the replay runs on the library's own stream, behind everything the caller has enqueued and in front of what it enqueues next
        const size_t nb = (size_t)T * nx * sizeof(double);
        HIPCHK(c, hipEventRecord(c->ev_g0, st));
        HIPCHK(c, hipStreamWaitEvent(c->sG, c->ev_g0, 0));
        rc = sweep_begin(c, seed, c->sG);
        if (rc) return rc;
        HIPCHK(c, hipMemcpyAsync(c->d_refbuf, ref_dev, nb, hipMemcpyDeviceToDevice, c->sG));
        HIPCHK(c, hipGraphLaunch(c->graph_exec, c->sG));
        HIPCHK(c, hipMemcpyAsync(traj_dev, c->d_trajbuf, nb, hipMemcpyDeviceToDevice, c->sG));
        HIPCHK(c, hipEventRecord(c->ev_g1, c->sG));
        HIPCHK(c, hipStreamWaitEvent(st, c->ev_g1, 0));
        c->last_graph = 1;
        return PGAS_OK;
    }
    rc = sweep_begin(c, seed, st);
    if (rc) return rc;
    return sweep_body(c, ref_dev, traj_dev, chunk, st);
}

int pgas_get_traces(pgas_ctx* c, double** x_trace, int32_t** anc_trace, double** logw_last, double** logw_trace) {
    if (!c) return PGAS_E_ARG;
    if (!c->have_traces) FAIL(c, PGAS_E_STATE, "pgas_get_traces: no sweep has run");
    if ((x_trace || anc_trace) && !c->rs[PG_RB_X].contiguous)
        FAIL(c, PGAS_E_STATE, "pgas_get_traces: this context keeps its traces in row blocks (sharded, or PGAS_OPT_TRACE_BLOCK_BYTES): use pgas_trace_layout / pgas_trace_row");
    if (x_trace) *x_trace = (double*)c->rs[PG_RB_X].blk[0];
    if (anc_trace) *anc_trace = (int32_t*)c->rs[PG_RB_ANC].blk[0];
    if (logw_last) *logw_last = c->logw_last;
    if (logw_trace) *logw_trace = c->logw_trace;
    return PGAS_OK;
}

/* Block layout of trace `kind` (PGAS_TRACE_*): info4 = {rows, rows per block (a power of two), blocks, bytes per row}.
 * Row t lives in block t / rows_per_block at row t % rows_per_block; pgas_trace_row returns its device pointer. */
int pgas_trace_layout(pgas_ctx* c, int32_t kind, int64_t* info4) {
    if (!c) return PGAS_E_ARG;
    if (!c->have_traces) FAIL(c, PGAS_E_STATE, "pgas_trace_layout: no traces yet (run a sweep or pgas_shard_setup first)");
    if (kind < 0 || kind >= PG_RB_NKIND || !info4) FAIL(c, PGAS_E_ARG, "pgas_trace_layout: bad argument");
    const RowStore& s = c->rs[kind];
    info4[0] = s.rows;
    info4[1] = s.contiguous ? s.rows : ((int64_t)1 << s.shift);
    info4[2] = s.nblk;
    info4[3] = (int64_t)s.row_bytes;
    return PGAS_OK;
}

int pgas_trace_row(pgas_ctx* c, int32_t kind, int32_t t, void** ptr) {
    if (!c) return PGAS_E_ARG;
    if (!c->have_traces) FAIL(c, PGAS_E_STATE, "pgas_trace_row: no traces yet");
    if (kind < 0 || kind >= PG_RB_NKIND || !ptr || t < 0 || t >= c->rs[kind < 0 || kind >= PG_RB_NKIND ? 0 : kind].rows)
        FAIL(c, PGAS_E_ARG, "pgas_trace_row: bad argument");
    *ptr = c->rs[kind].row(t);
    return PGAS_OK;
}

int pgas_last_final_index(pgas_ctx* c, int64_t* idx, void* stream) {
    if (!c) return PGAS_E_ARG;
    if (!idx) FAIL(c, PGAS_E_ARG, "pgas_last_final_index: NULL argument");
    if (!c->have_traces) FAIL(c, PGAS_E_STATE, "pgas_last_final_index: no sweep has run");
    DeviceGuard guard(c->device);
    HIPCHK(c, hipStreamSynchronize((hipStream_t)stream));
    UpperHdr h;
    HIPCHK(c, hipMemcpy(&h, c->sb[c->md.T & 1].hdr, sizeof h, hipMemcpyDeviceToHost));
    *idx = h.final_idx;
    return PGAS_OK;
}

int pgas_set_profiling(pgas_ctx* c, int32_t on) {
    if (!c) return PGAS_E_ARG;
    c->profiling = on ? 1 : 0;
    if (on > 1) c->prof_stride = on;  // on = n > 1: time every n-th launch
    if (on < 0) c->prof_stride = 1;   // on < 0: time every launch
    return PGAS_OK;
}

int pgas_set_option(pgas_ctx* c, int32_t option, int64_t value) {
    if (!c) return PGAS_E_ARG;
    if (option == PGAS_OPT_PROPAGATE_CHUNK) {
        if (value < 0) FAIL(c, PGAS_E_ARG, "pgas_set_option: chunk must be >= 0");
        c->prop_chunk = (int)value;
        return PGAS_OK;
    }
    if (option == PGAS_OPT_FORCE_SLOW_RESAMPLE) {
        c->force_slow = value ? 1 : 0;
        return PGAS_OK;
    }
    if (option == PGAS_OPT_TAIL_GROUPS) {
        c->tail_groups = value ? 1 : 0;
        return PGAS_OK;
    }
    if (option == PGAS_OPT_MAX_LEAD) {
        if (value < 0) FAIL(c, PGAS_E_ARG, "pgas_set_option: max lead must be >= 0");
        c->max_lead = (int)value;
        return PGAS_OK;
    }
    if (option == PGAS_OPT_SYRK_SPLITS) {
        if (value < 0 || value > 32) FAIL(c, PGAS_E_ARG, "pgas_set_option: SYRK splits must be 0 (automatic) .. 32");
        c->syrk_splits = (int)value;
        return PGAS_OK;
    }
    if (option == PGAS_OPT_EVENT_STRIDE) {
        if (value < 1) FAIL(c, PGAS_E_ARG, "pgas_set_option: event stride must be >= 1");
        c->ev_stride = (int)value;
        return PGAS_OK;
    }
    if (option == PGAS_OPT_LOCAL_GROUPS) {
        c->local_groups = value ? 1 : 0;
        return PGAS_OK;
    }
    if (option == PGAS_OPT_OVERLAP) {
        c->overlap = value ? 1 : 0;
        return PGAS_OK;
    }
    if (option == PGAS_OPT_PROPAGATE_LDS) {
        if (value < 0 || value > 160 * 1024) FAIL(c, PGAS_E_ARG, "pgas_set_option: LDS bytes out of range");
        c->prop_lds = (int)value;
        return PGAS_OK;
    }
    if (option == PGAS_OPT_MFMA_PROPAGATE) {
        if (value < 0 || value > 1) FAIL(c, PGAS_E_ARG, "pgas_set_option: PGAS_OPT_MFMA_PROPAGATE takes 0 or 1");
        c->use_mx = (int)value;
        return PGAS_OK;
    }
    if (option == PGAS_OPT_SMALL_SWEEP) {
        if (value < 0 || value > 2) FAIL(c, PGAS_E_ARG, "pgas_set_option: PGAS_OPT_SMALL_SWEEP takes 0, 1 or 2");
        c->use_small = (int)value;
        return PGAS_OK;
    }
    if (option == PGAS_OPT_GRAPH) {
        c->use_graph = value ? 1 : 0;
        return PGAS_OK;
    }
    if (option == PGAS_OPT_TRACE_BLOCK_BYTES) {
        if (value < 0) FAIL(c, PGAS_E_ARG, "pgas_set_option: block bytes must be >= 0");
        if (c->have_traces) FAIL(c, PGAS_E_STATE, "pgas_set_option: the traces are already allocated");
        c->trace_block_bytes = (size_t)value;
        return PGAS_OK;
    }
    if (option == PGAS_OPT_MNIW_VALU) {
        c->mniw_valu = value == 2 ? 2 : (value ? 1 : 0);   // 2: the two-rows-per-lane kernels of 63 <= M <= 126 at every M (test knob)
        return PGAS_OK;
    }
    if (option == PGAS_OPT_RESAMPLE_BEFORE_PROPAGATE) {
        if (value && c->sharded) FAIL(c, PGAS_E_STATE, "pgas_set_option: the corrected mode is not available on a sharded context");
        c->corrected = value ? 1 : 0;
        return PGAS_OK;
    }
    FAIL(c, PGAS_E_ARG, "pgas_set_option: unknown option %d", option);
}

int pgas_get_profile(pgas_ctx* c, int64_t* launches, double* total_ms, int64_t* propagate_launches, double* propagate_ms, void* stream) {
    if (!c) return PGAS_E_ARG;
    if (!launches || !total_ms || !propagate_launches || !propagate_ms) FAIL(c, PGAS_E_ARG, "pgas_get_profile: NULL argument");
    DeviceGuard guard(c->device);
    HIPCHK(c, hipStreamSynchronize((hipStream_t)stream));
    double sum = 0.0;
    for (int i = 0; i + 1 < c->ev_used; i += 2) {
        float ms = 0.f;
        HIPCHK(c, hipEventElapsedTime(&ms, c->ev[i], c->ev[i + 1]));
        sum += ms;
    }
    *launches = c->ev_used / 2;
    *total_ms = sum;
    double psum = 0.0;
    for (int i = 0; i + 1 < c->evp_used; i += 2) {
        float ms = 0.f;
        HIPCHK(c, hipEventElapsedTime(&ms, c->evp[i], c->evp[i + 1]));
        psum += ms;
    }
    *propagate_launches = c->evp_used / 2;
    *propagate_ms = psum;
    return PGAS_OK;
}

/* What the last sweep actually launched (bench.py labels its roofline block from this, not from constants):
 * info[0] = time steps per k_propagate launch, info[1] = 1 when k_step scanned the groups itself (single device, <= 1024
 * segments), 0 when k_groups ran between the steps, info[2] = padded innermost basis extent JP of the k_propagate variant,
 * info[3] = particles per basis pass P of that variant. */
int pgas_get_launch_info(pgas_ctx* c, int32_t* info4) {
    if (!c) return PGAS_E_ARG;
    if (!info4) FAIL(c, PGAS_E_ARG, "pgas_get_launch_info: NULL argument");
    info4[0] = c->last_chunk;
    info4[1] = (c->last_small ? 3 : (sweep_is_local(c) ? 1 : (sweep_tail_groups(c) ? 2 : 0))) + (c->last_graph ? 16 : 0) + ((c->d_mxi && c->use_mx && !c->last_small) ? 32 : 0);   // + 16: the last sweep replayed the captured graph; + 32: k_propagate ran its matrix-core form
    info4[2] = c->md.JP;
    info4[3] = c->var_P;
    return PGAS_OK;
}

#ifdef PG_DEBUG_DUMP
int pgas_debug_dump(double* out /* 4*64 */) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dbg), sizeof(double) * 4 * 64) == hipSuccess ? 0 : -2;
}
#endif
#ifdef PG_STAMPS
int pgas_debug_stamps(unsigned long long* out /* 2048*16 */) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 2048 * 16) == hipSuccess ? 0 : -2;
}
#endif


/* systematic_SISR (src/Filtering.py:6-37) on log-weights: idx[i] = first k with W_k >= (u + i)/N, W the canonical CDF of
 * softmax(logw).  Uses the context's scan scratch; logw_dev (N), idx_dev (N) int32. */
static int systematic_impl(pgas_ctx* c, double u, const double* u_dev, const double* logw_dev, int32_t* idx_dev, void* stream) {
    if (c->sharded) FAIL(c, PGAS_E_STATE, "pgas_systematic_resample: not available on a shard context");
    DeviceGuard guard(c->device);
    hipStream_t st = (hipStream_t)stream;
    const DevModel& md = c->md;
    install_own_peers(c);
    hipLaunchKernelGGL(k_segscan, dim3(md.nseg), dim3(PG_BLK), 0, st, md.N, logw_dev, c->sb[0]);
    KCHK(c, "k_segscan");
    int rc = launch_groups(c, c->sb[0], 1, st);
    if (rc) return rc;
    hipLaunchKernelGGL(k_systematic, dim3(md.nseg), dim3(PG_BLK), 0, st, md, u, u_dev, c->sb[0], peers_for(c, 0), idx_dev);
    KCHK(c, "k_systematic");
    return PGAS_OK;
}

int pgas_systematic_resample(pgas_ctx* c, double u, const double* logw_dev, int32_t* idx_dev, void* stream) {
    if (!c) return PGAS_E_ARG;
    if (!logw_dev || !idx_dev || !(u >= 0.0 && u < 1.0)) FAIL(c, PGAS_E_ARG, "pgas_systematic_resample: bad argument");
    return systematic_impl(c, u, nullptr, logw_dev, idx_dev, stream);
}

int pgas_systematic_resample_dev(pgas_ctx* c, const double* u_dev, const double* logw_dev, int32_t* idx_dev, void* stream) {
    if (!c) return PGAS_E_ARG;
    if (!u_dev || !logw_dev || !idx_dev) FAIL(c, PGAS_E_ARG, "pgas_systematic_resample_dev: NULL argument");
    return systematic_impl(c, 0.0, u_dev, logw_dev, idx_dev, stream);
}

/* reconstruct_trajectory (src/Filtering.py:40-55): x_dev (T,N,nx), anc_dev (T-1,N) int32 with anc[i][j] = index at time i of the
 * ancestor of particle j of time i+1, idx = index at time T-1; traj_dev (T,nx) out.  T and nx are passed explicitly, N is the context's. */
int pgas_reconstruct_trajectory(pgas_ctx* c, const double* x_dev, const int32_t* anc_dev, int32_t T, int32_t nx, int64_t idx,
                                double* traj_dev, void* stream) {
    if (!c) return PGAS_E_ARG;
    if (!x_dev || !traj_dev || T < 1 || nx < 1 || idx < 0 || idx >= c->md.N || (T > 1 && !anc_dev))
        FAIL(c, PGAS_E_ARG, "pgas_reconstruct_trajectory: bad argument");
    DeviceGuard guard(c->device);
    hipLaunchKernelGGL(k_backtrace_idx, dim3(1), dim3(64), 0, (hipStream_t)stream, c->md.N, T, nx, x_dev, anc_dev, idx, traj_dev);
    KCHK(c, "k_backtrace_idx");
    return PGAS_OK;
}

// ------------------------------------------------------------------------------------------------
// Particle-sharded sweep (one context per rank, each owning N_local = N_global / world particles).
// Per time step: k_step (local search + scans), ONE collective -- an all-gather of the per-segment softmax partials --,
// k_groups over the gathered partials (every rank computes the same group records); peers' cumsums, log-likelihood rows
// and traces are read directly through xGMI peer mappings whose pointers are installed with pgas_shard_set_peer.
// ------------------------------------------------------------------------------------------------
int pgas_shard_setup(pgas_ctx* c, int32_t rank, int32_t world) {
    if (!c) return PGAS_E_ARG;
    if (world < 1 || world > PG_MAX_RANKS || rank < 0 || rank >= world) FAIL(c, PGAS_E_ARG, "pgas_shard_setup: bad rank/world %d/%d", rank, world);
    if (c->md.N % PGAS_SEG) FAIL(c, PGAS_E_ARG, "pgas_shard_setup: local particle count %d must be a multiple of %d", c->md.N, PGAS_SEG);
    if ((int64_t)c->md.nseg * world > PG_MAX_NSEG) FAIL(c, PGAS_E_ARG, "pgas_shard_setup: %d global segments exceed %d", c->md.nseg * world, PG_MAX_NSEG);
    if (c->sharded) FAIL(c, PGAS_E_STATE, "pgas_shard_setup: already set up");
    if (c->corrected) FAIL(c, PGAS_E_STATE, "pgas_shard_setup: the corrected mode is not available on a sharded context");
    DeviceGuard guard(c->device);
    if (c->have_traces && c->rs[PG_RB_X].contiguous) {
        // an unsharded sweep ran on this context before: its traces are single allocations; a shard's are row blocks (IPC export)
        HIPCHK(c, hipDeviceSynchronize());
        for (RowStore& r : c->rs) r.release();
        hipFree(c->logw_last); hipFree(c->logw_trace);
        c->logw_last = c->logw_trace = nullptr;
        c->have_traces = false;
    }
    int rc = ensure_traces(c, true);
    if (rc) return rc;
    // allocate everything first, commit the context's state only when nothing can fail any more
    const int nsegp = c->sb[0].nsegp, nseg_g = c->md.nseg * world, nsegp_g = (nseg_g + 63) / 64 * 64;
    double* segk_g[2] = {nullptr, nullptr};
    uint64_t* segs_g[2] = {nullptr, nullptr};
    double* tab[2][3] = {{nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}};
    hipError_t e = hipSuccess;
    for (int i = 0; i < 2 && e == hipSuccess; ++i) {
        e = hipMalloc(&segk_g[i], (size_t)world * 2 * nsegp * sizeof(double));
        if (e == hipSuccess) e = hipMalloc(&segs_g[i], (size_t)world * 2 * nsegp * sizeof(uint64_t));
        for (int k = 0; k < 3 && e == hipSuccess; ++k) e = hipMalloc(&tab[i][k], 2 * (size_t)nsegp_g * sizeof(double));
        if (e == hipSuccess) e = hipMemset(segk_g[i], 0, (size_t)world * 2 * nsegp * sizeof(double));
        if (e == hipSuccess) e = hipMemset(segs_g[i], 0, (size_t)world * 2 * nsegp * sizeof(uint64_t));
    }
    if (e != hipSuccess) {
        for (int i = 0; i < 2; ++i) {
            hipFree(segk_g[i]); hipFree(segs_g[i]);
            for (int k = 0; k < 3; ++k) hipFree(tab[i][k]);
        }
        FAIL(c, e == hipErrorOutOfMemory ? PGAS_E_NOMEM : PGAS_E_HIP, "pgas_shard_setup: allocation failed: %s", hipGetErrorString(e));
    }
    DevModel& md = c->md;
    c->rank = rank; c->world = world;
    md.p0 = (int64_t)rank * md.N;
    md.Ng = md.N * world;
    md.nseg_g = nseg_g;
    for (int r = 0; r < PG_MAX_RANKS; ++r) {   // forget the single-device table: every rank must be installed again
        for (int k = 0; k < PG_RB_NKIND; ++k) c->peer_blk[r][k].clear();
        c->peer_c1[0][r] = c->peer_c1[1][r] = c->peer_c2[0][r] = c->peer_c2[1][r] = nullptr;
    }
    c->peers.world = world; c->peers.nseg_l = md.nseg; c->peers.Nl = md.N;
    for (int i = 0; i < 2; ++i) {
        ScanBufs& sb = c->sb[i];
        c->segk_g[i] = segk_g[i]; c->segs_g[i] = segs_g[i];
        sb.segk = segk_g[i]; sb.segs = segs_g[i];          // group scans read the gathered (world, 2, nsegp); local scans keep writing segk_w/segs_w
        sb.nseg_l = md.nseg; sb.rank_stride = 2 * nsegp; sb.nsegp_g = nsegp_g;
        hipFree(sb.tab_e); hipFree(sb.tab_sc); hipFree(sb.tab_m);
        sb.tab_e = tab[i][0]; sb.tab_sc = tab[i][1]; sb.tab_m = tab[i][2];
    }
    c->sharded = true;
    install_own_peers(c);
    return PGAS_OK;
}

/* out[17]: c1[0], c1[1], then the FIRST block of la, h, ln, x, anc (the seven buffers peers read; every block of them is listed by
 *          pgas_shard_layout / pgas_shard_block), c2[0], c2[1] (unused by the sweep), segk_w[0], segk_w[1], segs_w[0], segs_w[1],
 *          segk_g[0], segk_g[1], segs_g[0], segs_g[1];
 * sizes[3]: nsegp (local padded segments), N_local, T */
int pgas_shard_buffers(pgas_ctx* c, void** out, int64_t* sizes) {
    if (!c) return PGAS_E_ARG;
    if (!c->sharded) FAIL(c, PGAS_E_STATE, "pgas_shard_buffers: call pgas_shard_setup first");
    if (!out || !sizes) FAIL(c, PGAS_E_ARG, "pgas_shard_buffers: NULL argument");
    out[0] = c->sb[0].c1; out[1] = c->sb[1].c1;
    out[2] = c->rs[PG_RB_LA].blk[0]; out[3] = c->rs[PG_RB_H].blk[0]; out[4] = c->rs[PG_RB_LN].blk[0]; out[5] = c->rs[PG_RB_X].blk[0]; out[6] = c->rs[PG_RB_ANC].blk[0];
    out[7] = c->sb[0].c2; out[8] = c->sb[1].c2;
    out[9] = c->sb[0].segk_w; out[10] = c->sb[1].segk_w; out[11] = c->sb[0].segs_w; out[12] = c->sb[1].segs_w;
    out[13] = c->segk_g[0]; out[14] = c->segk_g[1]; out[15] = c->segs_g[0]; out[16] = c->segs_g[1];
    sizes[0] = c->sb[0].nsegp; sizes[1] = c->md.N; sizes[2] = c->md.T;
    return PGAS_OK;
}

// peer-visible buffer `which` (0, 1: the cumsum of scan buffer 0 / 1; 2..6: la, h, ln, x, anc) -> trace kind, or -1 for the cumsums
static int shard_which_kind(int which) {
    static const int k[7] = {-1, -1, PG_RB_LA, PG_RB_H, PG_RB_LN, PG_RB_X, PG_RB_ANC};
    return k[which];
}

/* Blocks of peer-visible buffer `which` (0..6 as in pgas_shard_buffers): info2 = {number of blocks, rows per block}.  The cumsums are
 * one block each; the traces are row blocks of at most 1 GiB (PGAS_OPT_TRACE_BLOCK_BYTES), the same layout on every rank. */
int pgas_shard_layout(pgas_ctx* c, int32_t which, int64_t* info2) {
    if (!c) return PGAS_E_ARG;
    if (!c->sharded) FAIL(c, PGAS_E_STATE, "pgas_shard_layout: call pgas_shard_setup first");
    if (which < 0 || which > 6 || !info2) FAIL(c, PGAS_E_ARG, "pgas_shard_layout: bad argument");
    const int k = shard_which_kind(which);
    info2[0] = k < 0 ? 1 : c->rs[k].nblk;
    info2[1] = k < 0 ? 1 : ((int64_t)1 << c->rs[k].shift);
    return PGAS_OK;
}

/* Device pointer and size of block `blk` of this rank's buffer `which`. */
int pgas_shard_block(pgas_ctx* c, int32_t which, int32_t blk, void** ptr, int64_t* bytes) {
    if (!c) return PGAS_E_ARG;
    if (!c->sharded) FAIL(c, PGAS_E_STATE, "pgas_shard_block: call pgas_shard_setup first");
    if (which < 0 || which > 6 || !ptr) FAIL(c, PGAS_E_ARG, "pgas_shard_block: bad argument");
    const int k = shard_which_kind(which);
    if (blk < 0 || blk >= (k < 0 ? 1 : c->rs[k].nblk)) FAIL(c, PGAS_E_ARG, "pgas_shard_block: buffer %d has no block %d", which, blk);
    if (k < 0) {
        *ptr = c->sb[which].c1;
        if (bytes) *bytes = (int64_t)((size_t)c->md.nseg * PGAS_SEG * sizeof(uint64_t));
    } else {
        *ptr = c->rs[k].blk[blk];
        if (bytes) *bytes = (int64_t)((size_t)c->rs[k].rows_in_block(blk) * c->rs[k].row_bytes);
    }
    return PGAS_OK;
}

/* Block `blk` of rank `peer`'s buffer `which` as mapped into THIS process (same process: the pointer itself; another process:
 * what pgas_ipc_open returned for the handle of pgas_ipc_export(which, blk) on that rank). */
int pgas_shard_set_peer_block(pgas_ctx* c, int32_t peer, int32_t which, int32_t blk, const void* ptr) {
    if (!c) return PGAS_E_ARG;
    if (!c->sharded) FAIL(c, PGAS_E_STATE, "pgas_shard_set_peer_block: call pgas_shard_setup first");
    if (peer < 0 || peer >= c->world || which < 0 || which > 6 || !ptr) FAIL(c, PGAS_E_ARG, "pgas_shard_set_peer_block: bad argument");
    const int k = shard_which_kind(which);
    if (blk < 0 || blk >= (k < 0 ? 1 : c->rs[k].nblk)) FAIL(c, PGAS_E_ARG, "pgas_shard_set_peer_block: buffer %d has no block %d", which, blk);
    if (k < 0) {
        c->peer_c1[which][peer] = (const uint64_t*)ptr;
    } else {
        auto& v = c->peer_blk[peer][k];
        if ((int)v.size() != c->rs[k].nblk) v.assign((size_t)c->rs[k].nblk, nullptr);
        v[blk] = (const char*)ptr;
        c->bt_dirty = true;
    }
    return PGAS_OK;
}

static int shard_ready(pgas_ctx* c, const char* who) {
    if (!c->sharded) FAIL(c, PGAS_E_STATE, "%s: call pgas_shard_setup first", who);
    if (!c->have_params) FAIL(c, PGAS_E_STATE, "%s: call pgas_set_params first", who);
    for (int r = 0; r < c->world; ++r) {
        bool ok = c->peer_c1[0][r] && c->peer_c1[1][r];
        for (int k = 0; k < PG_RB_NKIND && ok; ++k) {
            ok = (int)c->peer_blk[r][k].size() == c->rs[k].nblk;
            for (size_t b = 0; ok && b < c->peer_blk[r][k].size(); ++b) ok = c->peer_blk[r][k][b] != nullptr;
        }
        if (!ok) FAIL(c, PGAS_E_STATE, "%s: buffers of rank %d not installed (pgas_shard_set_peer_block)", who, r);
    }
    return PGAS_OK;
}

int pgas_shard_run(pgas_ctx* c, int32_t phase, int32_t t, int32_t t_aux, uint64_t seed, const double* ref_dev, double* traj_dev, void* stream) {
    if (!c) return PGAS_E_ARG;
    int rc = shard_ready(c, "pgas_shard_run");
    if (rc) return rc;
    DeviceGuard guard(c->device);
    hipStream_t st = (hipStream_t)stream;
    const DevModel& md = c->md;
    const int N = md.N, T = md.T;
    const dim3 grid(md.nseg), blk(PG_BLK);
    switch (phase) {
    case PGAS_SHARD_INIT:
        if (!ref_dev) FAIL(c, PGAS_E_ARG, "pgas_shard_run(INIT): ref_dev == NULL");
        rc = sweep_begin(c, seed, st);
        if (rc) return rc;
        hipLaunchKernelGGL(c->init, dim3((N + PG_BLK - 1) / PG_BLK), blk, 0, st, md, seed, (const SweepParams*)nullptr, c->d_m0L0, ref_dev, (double*)c->rs[PG_RB_X].row(0));
        KCHK(c, "k_init");
        return PGAS_OK;
    case PGAS_SHARD_PROPAGATE:  // time steps [t, t_aux)
        if (!ref_dev || t < 1 || t_aux > T || t >= t_aux) FAIL(c, PGAS_E_ARG, "pgas_shard_run(PROPAGATE): bad range [%d,%d)", t, t_aux);
        return launch_propagate(c, t, t_aux, ref_dev, st, false);
    case PGAS_SHARD_STEP:  // launch t in [1, T]: resample step t-1 (t > 1), scan step t (t < T)
        if (t < 1 || t > T) FAIL(c, PGAS_E_ARG, "pgas_shard_run(STEP): t = %d outside [1, %d]", t, T);
        return launch_step(c, t, false, st, nullptr, nullptr);
    case PGAS_SHARD_GROUPS:  // after the all-gather of step t's partials
        if (t < 1 || t >= T) FAIL(c, PGAS_E_ARG, "pgas_shard_run(GROUPS): t = %d outside [1, %d)", t, T);
        return launch_groups(c, c->sb[t & 1], 2, st);
    case PGAS_SHARD_FINAL_SCAN:
        hipLaunchKernelGGL(k_segscan, grid, blk, 0, st, N, c->logw_last, c->sb[T & 1]);
        KCHK(c, "k_segscan");
        return PGAS_OK;
    case PGAS_SHARD_FINAL:  // after the all-gather of the final scan's partials: final index (src/PGAS.py:224-225)
        rc = launch_groups(c, c->sb[T & 1], 1, st);
        if (rc) return rc;
        return launch_count(c, c->sb[T & 1], T & 1, 1, 0.0, st);
    case PGAS_SHARD_BACKTRACE:
        if (!traj_dev) FAIL(c, PGAS_E_ARG, "pgas_shard_run(BACKTRACE): traj_dev == NULL");
        return launch_backtrace(c, c->sb[T & 1].hdr, traj_dev, st);
    default:
        FAIL(c, PGAS_E_ARG, "pgas_shard_run: unknown phase %d", phase);
    }
}

#define NCCLCHK(c, call)                                                                                                   \
    do {                                                                                                                   \
        ncclResult_t r_ = (call);                                                                                          \
        if (r_ != ncclSuccess) FAIL(c, PGAS_E_HIP, "%s failed: %s", #call, rccl().error_string ? rccl().error_string(r_) : "RCCL error"); \
    } while (0)

int pgas_shard_unique_id(void* id128) {
    if (!id128) return PGAS_E_ARG;
    if (!rccl().ok) return PGAS_E_STATE;
    return rccl().get_unique_id((ncclUniqueId*)id128) == ncclSuccess ? PGAS_OK : PGAS_E_HIP;
}

int pgas_shard_comm_init(pgas_ctx* c, const void* id128) {
    if (!c) return PGAS_E_ARG;
    if (!c->sharded) FAIL(c, PGAS_E_STATE, "pgas_shard_comm_init: call pgas_shard_setup first");
    if (!id128) FAIL(c, PGAS_E_ARG, "pgas_shard_comm_init: NULL id");
    if (!rccl().ok) FAIL(c, PGAS_E_STATE, "pgas_shard_comm_init: librccl.so could not be loaded");
    if (c->comm) return PGAS_OK;
    DeviceGuard guard(c->device);
    ncclUniqueId id;
    memcpy(&id, id128, sizeof id);
    NCCLCHK(c, rccl().init_rank(&c->comm, c->world, id, c->rank));
    HIPCHK(c, hipMalloc(&c->d_sync, (size_t)(c->world + 1) * sizeof(int32_t)));
    HIPCHK(c, hipMemset(c->d_sync, 0, (size_t)(c->world + 1) * sizeof(int32_t)));
    return PGAS_OK;
}

int pgas_shard_set_collective(pgas_ctx* c, pgas_allgather_fn fn, void* user) {
    if (!c) return PGAS_E_ARG;
    if (!c->sharded) FAIL(c, PGAS_E_STATE, "pgas_shard_set_collective: call pgas_shard_setup first");
    c->ag_cb = fn;
    c->ag_user = user;
    return PGAS_OK;
}

static int shard_all_gather(pgas_ctx* c, int parity, hipStream_t st) {
    const size_t cnt = 2 * (size_t)c->sb[parity].nsegp;
    if (c->ag_cb) {
        // host-staged collective (tests: several ranks on one device, where RCCL refuses duplicate devices).  parity < 0 asks for
        // the end-of-sweep barrier.  The callback synchronises the stream itself.
        const int rc = c->ag_cb(c->ag_user, parity, (void*)st);
        if (rc) FAIL(c, PGAS_E_HIP, "all-gather callback failed (%d)", rc);
        return PGAS_OK;
    }
    if (!c->comm) FAIL(c, PGAS_E_STATE, "sharded sweep: call pgas_shard_comm_init (RCCL) or pgas_shard_set_collective first");
    if (parity < 0) {
        NCCLCHK(c, rccl().all_gather(c->d_sync + c->world, c->d_sync, 1, ncclInt32, c->comm, st));
        return PGAS_OK;
    }
    NCCLCHK(c, rccl().group_start());
    NCCLCHK(c, rccl().all_gather(c->sb[parity].segk_w, c->segk_g[parity], cnt, ncclDouble, c->comm, st));
    NCCLCHK(c, rccl().all_gather(c->sb[parity].segs_w, c->segs_g[parity], cnt, ncclUint64, c->comm, st));
    NCCLCHK(c, rccl().group_end());
    return PGAS_OK;
}

int pgas_shard_sweep(pgas_ctx* c, uint64_t seed, const double* ref_dev, double* traj_dev, int32_t propagate_chunk, void* stream) {
    if (!c) return PGAS_E_ARG;
    int rc = shard_ready(c, "pgas_shard_sweep");
    if (rc) return rc;
    if (!c->comm && !c->ag_cb) FAIL(c, PGAS_E_STATE, "pgas_shard_sweep: call pgas_shard_comm_init (RCCL) or pgas_shard_set_collective first");
    if (!ref_dev || !traj_dev) FAIL(c, PGAS_E_ARG, "pgas_shard_sweep: NULL argument");
    DeviceGuard guard(c->device);
    hipStream_t st = (hipStream_t)stream;
    const DevModel& md = c->md;
    const int T = md.T;
    rc = sweep_begin(c, seed, st);
    if (rc) return rc;
    hipLaunchKernelGGL(c->init, dim3((md.N + PG_BLK - 1) / PG_BLK), dim3(PG_BLK), 0, st, md, seed, (const SweepParams*)nullptr, c->d_m0L0, ref_dev, (double*)c->rs[PG_RB_X].row(0));
    KCHK(c, "k_init");
    c->ev_used = 0;
    c->evp_used = 0;
    if (T == 1) {
        HIPCHK(c, hipMemsetAsync(c->logw_last, 0, md.N * sizeof(double), st));
    } else {
        const int chunk = propagate_chunk > 0 ? propagate_chunk : (c->prop_chunk > 0 ? c->prop_chunk : (c->overlap ? 1 : T));
        c->last_chunk = chunk;
        rc = run_time_loop(c, ref_dev, chunk, st);
        if (rc) return rc;
    }
    rc = run_final(c, traj_dev, st);
    if (rc) return rc;
    // peers may still be chasing ancestors through this rank's traces: one more (tiny) collective closes the sweep on every rank
    return shard_all_gather(c, -1, st);
}

/* Measurement aid: `reps` back-to-back issues of the step's all-gather (scan buffer 0) on `stream`, for timing it in isolation.
 * Only between sweeps (it overwrites the gathered partials of scan buffer 0, which the next sweep rewrites anyway). */
int pgas_shard_probe_collective(pgas_ctx* c, int32_t reps, void* stream) {
    if (!c) return PGAS_E_ARG;
    if (!c->sharded || !c->comm) FAIL(c, PGAS_E_STATE, "pgas_shard_probe_collective: needs an RCCL communicator (pgas_shard_comm_init)");
    if (reps < 1) FAIL(c, PGAS_E_ARG, "pgas_shard_probe_collective: reps must be >= 1");
    DeviceGuard guard(c->device);
    pgas_allgather_fn keep = c->ag_cb;
    c->ag_cb = nullptr;
    int rc = PGAS_OK;
    for (int i = 0; i < reps && rc == PGAS_OK; ++i) rc = shard_all_gather(c, 0, (hipStream_t)stream);
    c->ag_cb = keep;
    return rc;
}

/* xGMI / IPC plumbing for peers in OTHER processes: export a 64-byte handle of one of this context's buffers
 * (index as in pgas_shard_buffers, 0..6) and map a peer's handle into this process. */
int32_t pgas_hip_runtime_version(void) {
    int v = 0;
    return hipRuntimeGetVersion(&v) == hipSuccess ? v : -1;
}

int pgas_ipc_export(pgas_ctx* c, int32_t which, int32_t blk, void* handle64) {
    if (!c) return PGAS_E_ARG;
    if (!handle64) FAIL(c, PGAS_E_ARG, "pgas_ipc_export: NULL argument");
    void* p = nullptr;
    int64_t bytes = 0;
    int rc = pgas_shard_block(c, which, blk, &p, &bytes);
    if (rc) return rc;
    // the blocks are sized to stay below this (pgas_shard_setup); a caller-chosen PGAS_OPT_TRACE_BLOCK_BYTES may not
    if (bytes >= ((int64_t)1 << 31)) FAIL(c, PGAS_E_ARG, "pgas_ipc_export: block of %lld bytes; blocks of 2 GiB and more cannot be mapped by every HIP runtime", (long long)bytes);
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "handle size");
    DeviceGuard guard(c->device);
    HIPCHK(c, hipIpcGetMemHandle((hipIpcMemHandle_t*)handle64, p));
    return PGAS_OK;
}
int pgas_ipc_open(pgas_ctx* c, const void* handle64, void** ptr) {
    if (!c) return PGAS_E_ARG;
    if (!handle64 || !ptr) FAIL(c, PGAS_E_ARG, "pgas_ipc_open: NULL argument");
    DeviceGuard guard(c->device);
    if (!c->peer_access_tried) {   // best effort, once: let this device's kernels reach every other device of the node (xGMI)
        c->peer_access_tried = true;
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) == hipSuccess)
            for (int d = 0; d < ndev; ++d) {
                int can = 0;
                if (d != c->device && hipDeviceCanAccessPeer(&can, c->device, d) == hipSuccess && can) (void)hipDeviceEnablePeerAccess(d, 0);
            }
        (void)hipGetLastError();   // "already enabled" is not an error here
    }
    hipIpcMemHandle_t h;
    memcpy(&h, handle64, sizeof h);
    HIPCHK(c, hipIpcOpenMemHandle(ptr, h, hipIpcMemLazyEnablePeerAccess));
    return PGAS_OK;
}

int pgas_detmath_eval(int32_t device, int32_t which, const double* x_dev, const double* y_dev, const uint32_t* w_dev, int64_t n, double* out0_dev,
                      double* out1_dev, uint32_t* outw_dev, void* stream) {
    if (which < 0 || which > 7 || n < 0) return PGAS_E_ARG;
    const bool words = which == 3 || which == 7;
    if (words ? !w_dev : !x_dev) return PGAS_E_ARG;
    if ((which == 5 || which == 6) && !y_dev) return PGAS_E_ARG;
    if (which == 3 ? !outw_dev : !out0_dev) return PGAS_E_ARG;
    if ((which == 2 || which == 7) && !out1_dev) return PGAS_E_ARG;
    if (n == 0) return PGAS_OK;
    DeviceGuard guard(device);
    hipLaunchKernelGGL(k_detmath, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, which, x_dev, y_dev, w_dev, n, out0_dev, out1_dev, outw_dev);
    return hipGetLastError() == hipSuccess ? PGAS_OK : PGAS_E_HIP;
}

int pgas_suffstats(pgas_ctx* c, const double* traj_dev, double* T0_dev, double* T1_dev, double* T2_dev, void* stream) {
    if (!c) return PGAS_E_ARG;
    if (!traj_dev || !T0_dev || !T1_dev || !T2_dev) FAIL(c, PGAS_E_ARG, "pgas_suffstats: NULL argument");
    const DevModel& md = c->md;
    if (md.T < 2) FAIL(c, PGAS_E_ARG, "pgas_suffstats: needs T >= 2");
    DeviceGuard guard(c->device);
    hipStream_t st = (hipStream_t)stream;
    const int R = md.T - 1;                         // rows: t = 0..T-2   (traj[:-1], inputs[:-1]; Q3)
    const int Mp = (md.M + md.nx + SY_BM - 1) / SY_BM * SY_BM;   // [Phi | X+] padded to the 64-wide block
    const int Rp = (R + SY_KB - 1) / SY_KB * SY_KB;
    const int nb = Mp / SY_BM, ntri = nb * (nb + 1) / 2, nkb = Rp / SY_KB;
    int S = c->syrk_splits > 0 ? c->syrk_splits : 1024 / ntri;   // automatic: the largest split that keeps ntri x S within four workgroups per CU
    S = std::max(1, std::min(std::min(S, 32), nkb));
    const int kb_per_split = (nkb + S - 1) / S;
    S = (nkb + kb_per_split - 1) / kb_per_split;
    if (!c->d_phi) HIPCHK(c, hipMalloc(&c->d_phi, (size_t)Rp * Mp * sizeof(double)));
    const size_t ws_need = (size_t)S * ntri * SY_BM * SY_BM * sizeof(double);
    if (ws_need > c->syrk_ws_bytes) {
        if (c->d_syrk_ws) HIPCHK(c, hipFree(c->d_syrk_ws));
        c->d_syrk_ws = nullptr; c->syrk_ws_bytes = 0;
        HIPCHK(c, hipMalloc(&c->d_syrk_ws, ws_need));
        c->syrk_ws_bytes = ws_need;
    }
    hipLaunchKernelGGL(md.nx == 1 ? k_traj_basis<1> : k_traj_basis<2>, dim3(Rp / SY_RB), dim3(256), 0, st, md, c->d_idx, traj_dev, R, Rp, Mp, c->d_phi);
    KCHK(c, "k_traj_basis");
    hipLaunchKernelGGL(k_syrk_lds, dim3(ntri, S), dim3(256), 0, st, c->d_phi, Rp, Mp, kb_per_split, c->d_syrk_ws);
    KCHK(c, "k_syrk_lds");
    hipLaunchKernelGGL(k_syrk_reduce, dim3(ntri, SY_BM * SY_BM / 256), dim3(256), 0, st, c->d_syrk_ws, ntri, S, md.M, md.nx, T0_dev, T1_dev, T2_dev);
    KCHK(c, "k_syrk_reduce");
    return PGAS_OK;
}

}  // extern "C"


// ------------------------------------------------------------------------------------------ marginalised family (pgas_marginal.h)
double pgas_m_rng_uniform(uint64_t seed, uint32_t stream, uint32_t t) { return pgas_rng_uniform(seed, stream, t); }

int pgas_m_rng_normal(pgas_ctx* c, uint64_t seed, uint32_t stream, uint32_t t, int64_t p0, int64_t n, int32_t ncol, double* out, void* sh) {
    if (!c) return PGAS_E_ARG;
    if (!out || n < 0 || ncol < 1 || ncol > 8) FAIL(c, PGAS_E_ARG, "pgas_m_rng_normal: bad argument (n = %lld, ncol = %d)", (long long)n, ncol);
    if (n == 0) return PGAS_OK;
    DeviceGuard guard(c->device);
    hipLaunchKernelGGL(k_rng_normal, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)sh, seed, stream, t, c->t_dev, p0, n, ncol, out);
    KCHK(c, "k_rng_normal");
    return PGAS_OK;
}

int pgas_m_set_time_source(pgas_ctx* c, const uint32_t* t_dev) {
    if (!c) return PGAS_E_ARG;
    c->t_dev = t_dev;
    return PGAS_OK;
}

int pgas_m_rng_uniform_dev(pgas_ctx* c, uint64_t seed, uint32_t stream, uint32_t t, double* out_dev, void* sh) {
    if (!c) return PGAS_E_ARG;
    if (!out_dev) FAIL(c, PGAS_E_ARG, "pgas_m_rng_uniform_dev: NULL argument");
    DeviceGuard guard(c->device);
    hipLaunchKernelGGL(k_rng_uniform_dev, dim3(1), dim3(64), 0, (hipStream_t)sh, seed, stream, t, c->t_dev, out_dev);
    KCHK(c, "k_rng_uniform_dev");
    return PGAS_OK;
}

int pgas_m_rng_student_t(pgas_ctx* c, uint64_t seed, uint32_t stream, uint32_t t, int64_t p0, int64_t n, const double* nu, double* out, void* sh) {
    if (!c) return PGAS_E_ARG;
    if (!out || !nu || n < 0) FAIL(c, PGAS_E_ARG, "pgas_m_rng_student_t: bad argument");
    if (n == 0) return PGAS_OK;
    DeviceGuard guard(c->device);
    hipLaunchKernelGGL(k_rng_student_t, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)sh, seed, stream, t, c->t_dev, p0, n, nu, out);
    KCHK(c, "k_rng_student_t");
    return PGAS_OK;
}

/* Student-t variates whose degrees of freedom are formed in the kernel: nu[p] = nu0 + nu_scale * src[anc[p]] (df = P3 + lambda T3[a], BI:45). */
int pgas_m_rng_student_t_df(pgas_ctx* c, uint64_t seed, uint32_t stream, uint32_t t, int64_t p0, int64_t n, const int32_t* anc, const double* src, double nu0,
                            double nu_scale, double* out, void* sh) {
    if (!c) return PGAS_E_ARG;
    if (!out || !src || !anc || n < 0) FAIL(c, PGAS_E_ARG, "pgas_m_rng_student_t_df: bad argument");
    if (n == 0) return PGAS_OK;
    DeviceGuard guard(c->device);
    hipLaunchKernelGGL(k_rng_student_t, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)sh, seed, stream, t, c->t_dev, p0, n, src, out, anc, nu0, nu_scale);
    KCHK(c, "k_rng_student_t");
    return PGAS_OK;
}

/* xi[p] = m[p] + sqrt((P2 + scale T2[a] - q[a]) / (P3 + scale T3[a])) t[p] sqrt(c[p] + 1), a = anc[p]: the scalar matrix-t draw (BI:64-108) */
int pgas_m_mniw_draw(pgas_ctx* c, int64_t n, double scale, const int32_t* anc, const double* m, const double* cc, const double* q, const double* T2,
                     const double* T3, double P2, double P3, const double* t, double* out, void* sh) {
    if (!c) return PGAS_E_ARG;
    if (!m || !cc || !q || !T2 || !T3 || !t || !out || n < 0) FAIL(c, PGAS_E_ARG, "pgas_m_mniw_draw: NULL argument");
    if (n == 0) return PGAS_OK;
    DeviceGuard guard(c->device);
    hipLaunchKernelGGL(k_mniw_draw, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)sh, n, scale, anc, m, cc, q, T2, T3, P2, P3, t, out);
    KCHK(c, "k_mniw_draw");
    return PGAS_OK;
}

/* g[p] = lbm(prior + T[p]) - lbm(prior + T[p] + R): the ancestor weights' base-measure term (src/Algorithm3.py:93-108), n = 1; r2 / r3 on the device */
int pgas_m_lbm_diff(pgas_ctx* c, int64_t n, int32_t M, const double* T2, const double* T3, const double* q1, const double* ld1, const double* q2,
                    const double* ld2, double P2, double P3, const double* r2_dev, const double* r3_dev, double* out, void* sh) {
    if (!c) return PGAS_E_ARG;
    if (!T2 || !T3 || !q1 || !ld1 || !q2 || !ld2 || !r2_dev || !r3_dev || !out || n < 0) FAIL(c, PGAS_E_ARG, "pgas_m_lbm_diff: NULL argument");
    if (n == 0) return PGAS_OK;
    DeviceGuard guard(c->device);
    hipLaunchKernelGGL(k_lbm_diff, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)sh, n, M, T2, T3, q1, ld1, q2, ld2, P2, P3, r2_dev, r3_dev, out);
    KCHK(c, "k_lbm_diff");
    return PGAS_OK;
}

/* phi (n, M) = the Hilbert basis at concat(state[p], input)[sel] / div for every particle (src/BasisFunctions.py:77-80): D <= 4 dimensions, the
 * per-dimension tables on the host, the (M, D) index table on the device. */
int pgas_m_hilbert_basis(pgas_ctx* c, int64_t n, int32_t M, int32_t D, const double* state, int32_t nx, const double* input, int32_t nu, const int32_t* sel,
                         const double* div, const double* center, const double* L, const double* size, const int32_t* idx_dev, double* out, void* sh) {
    if (!c) return PGAS_E_ARG;
    if (!state || !sel || !div || !center || !L || !size || !idx_dev || !out || n < 0) FAIL(c, PGAS_E_ARG, "pgas_m_hilbert_basis: NULL argument");
    if (D < 1 || D > PG_HB_MAXD || M < 1) FAIL(c, PGAS_E_ARG, "pgas_m_hilbert_basis: D = %d outside [1, %d]", D, PG_HB_MAXD);
    HilbertArgs h{};
    h.D = D; h.nx = nx; h.nu = nu;
    for (int d = 0; d < D; ++d) {
        if (sel[d] < 0 || sel[d] >= nx + nu || (sel[d] >= nx && !input)) FAIL(c, PGAS_E_ARG, "pgas_m_hilbert_basis: sel[%d] = %d outside the state / input", d, sel[d]);
        h.sel[d] = sel[d]; h.div[d] = div[d]; h.center[d] = center[d]; h.L[d] = L[d]; h.size[d] = size[d]; h.amp[d] = std::sqrt(1.0 / L[d]);
    }
    if (n == 0) return PGAS_OK;
    DeviceGuard guard(c->device);
    hipLaunchKernelGGL(k_hilbert_batch, dim3((unsigned)((n * M + 255) / 256)), dim3(256), 0, (hipStream_t)sh, n, M, h, state, input, idx_dev, out);
    KCHK(c, "k_hilbert_batch");
    return PGAS_OK;
}

/* The same variates on the HOST (no context, no device): the library's own arithmetic (include/pgas_canon.h) compiled for the CPU, so
 * that host-side helpers (pgas_amd.prior_mniw_drawPred, BI:92-108) draw from the SAME streams as the kernels, bit for bit. */
int pgas_m_rng_student_t_host(uint64_t seed, uint32_t stream, uint32_t t, int64_t p0, int64_t n, const double* nu_host, double* out_host) {
    if (!nu_host || !out_host || n < 0) return PGAS_E_ARG;
    for (int64_t p = 0; p < n; ++p) out_host[p] = pgas_rng_student_t(seed, stream, t, (uint64_t)(p0 + p), nu_host[p]);
    return PGAS_OK;
}

int pgas_m_rng_chi2(pgas_ctx* c, uint64_t seed, uint32_t stream, uint32_t t, int64_t p0, int64_t n, const double* nu, double* out, void* sh) {
    if (!c) return PGAS_E_ARG;
    if (!out || !nu || n < 0) FAIL(c, PGAS_E_ARG, "pgas_m_rng_chi2: bad argument");
    if (n == 0) return PGAS_OK;
    DeviceGuard guard(c->device);
    hipLaunchKernelGGL(k_rng_chi2, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)sh, seed, stream, t, p0, n, nu, out);
    KCHK(c, "k_rng_chi2");
    return PGAS_OK;
}

int pgas_m_mniw_solve(pgas_ctx* c, int64_t n, int32_t M, double scale, const int32_t* anc, const double* P0, const double* P1, const double* T0, const double* T1,
                      const double* R0, const double* R1, const double* phi, double* m, double* cc, double* q, double* logdet, double* Lfac, void* sh) {
    return pgas_m_mniw_solve_n(c, n, M, 1, scale, anc, P0, P1, T0, T1, R0, R1, phi, m, cc, q, logdet, Lfac, sh);
}

int pgas_m_mniw_solve_n(pgas_ctx* c, int64_t n, int32_t M, int32_t nv, double scale, const int32_t* anc, const double* P0, const double* P1, const double* T0,
                        const double* T1, const double* R0, const double* R1, const double* phi, double* m, double* cc, double* q, double* logdet, double* Lfac,
                        void* sh) {
    if (!c) return PGAS_E_ARG;
    if (!P0 || !P1 || !T0 || !T1 || n < 0) FAIL(c, PGAS_E_ARG, "pgas_m_mniw_solve: NULL argument");
    if (nv < 1 || nv > 8) FAIL(c, PGAS_E_ARG, "pgas_m_mniw_solve: %d components of the interface variable outside [1, 8]", nv);
    if (M < 1 || M + 1 + nv > PG_MN_MAXROWS_WIDE) FAIL(c, PGAS_E_ARG, "pgas_m_mniw_solve: M = %d outside [1, %d]", M, PG_MN_MAXROWS_WIDE - 1 - nv);
    if ((R0 == nullptr) != (R1 == nullptr)) FAIL(c, PGAS_E_ARG, "pgas_m_mniw_solve: R0 and R1 must be given together");
    if (n == 0) return PGAS_OK;
    DeviceGuard guard(c->device);
    hipStream_t st = (hipStream_t)sh;
    if (!c->d_fail) {
        HIPCHK(c, hipMalloc(&c->d_fail, sizeof(int32_t)));
        HIPCHK(c, hipMemsetAsync(c->d_fail, 0, sizeof(int32_t), st));
    }
    if (M > PG_MN_MAXM || nv > 1 || c->mniw_valu == 2) {
        // 63 ... 126 basis functions, several components (or the test knob): two rows per lane, the triangle in LDS
        const size_t lds = (size_t)((M + 1 + nv) * (M + 2 + nv) / 2) * sizeof(double);
        if (lds > 64 * 1024) HIPCHK(c, hipFuncSetAttribute((const void*)k_mniw_solve_wide, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k_mniw_solve_wide, dim3((unsigned)n), dim3(64), lds, st, n, M, nv, scale, anc, P0, P1, T0, T1, R0, R1, phi, m, cc, q, logdet, Lfac, c->d_fail);
    } else if (c->mniw_valu) {
        const int MT = M <= 22 ? 24 : M <= 30 ? 32 : M <= 42 ? 44 : M <= 46 ? 48 : 64;   // rows: M + 2 (the right-hand sides ride along)
        const int waves = MT == 64 ? 2 : 4;   // LDS: waves x MT (MT+1)/2 doubles <= 64 KB
        const dim3 grd((unsigned)((n + waves - 1) / waves)), blk(64 * waves);
        auto kern = MT == 24 ? k_mniw_solve<24> : MT == 32 ? k_mniw_solve<32> : MT == 44 ? k_mniw_solve<44> : MT == 48 ? k_mniw_solve<48> : k_mniw_solve<64>;
        hipLaunchKernelGGL(kern, grd, blk, (size_t)waves * (MT * (MT + 1) / 2) * sizeof(double), st, n, M, scale, anc, P0, P1, T0, T1, R0, R1, phi, m, cc, q,
                           logdet, Lfac, c->d_fail);
    } else {
        // default: panels of four columns, trailing update on the f64 matrix cores
        const int NT = (M + 2 + 15) / 16;
        const int R = 16 * NT, perw = 4 * R + R * (R + 1) / 2;
        const int waves = NT == 4 ? 3 : 4;   // LDS: waves x perw doubles <= 64 KB
        const dim3 grd((unsigned)((n + waves - 1) / waves)), blk(64 * waves);
        auto kern = NT == 1 ? k_mniw_solve_mfma<1> : NT == 2 ? k_mniw_solve_mfma<2> : NT == 3 ? k_mniw_solve_mfma<3> : k_mniw_solve_mfma<4>;
        hipLaunchKernelGGL(kern, grd, blk, (size_t)waves * perw * sizeof(double), st, n, M, scale, anc, P0, P1, T0, T1, R0, R1, phi, m, cc, q, logdet, Lfac,
                           c->d_fail);
    }
    KCHK(c, "k_mniw_solve");
    return PGAS_OK;
}

int pgas_m_mniw_trisolve(pgas_ctx* c, int64_t n, int32_t M, const int32_t* anc, const double* Lfac, const double* phi, double* m, double* cc, void* sh) {
    return pgas_m_mniw_trisolve_n(c, n, M, 1, anc, Lfac, phi, m, cc, sh);
}

int pgas_m_mniw_trisolve_n(pgas_ctx* c, int64_t n, int32_t M, int32_t nv, const int32_t* anc, const double* Lfac, const double* phi, double* m, double* cc, void* sh) {
    if (!c) return PGAS_E_ARG;
    if (!Lfac || !phi || n < 0) FAIL(c, PGAS_E_ARG, "pgas_m_mniw_trisolve: NULL argument");
    if (nv < 1 || nv > 8) FAIL(c, PGAS_E_ARG, "pgas_m_mniw_trisolve: %d components of the interface variable outside [1, 8]", nv);
    if (M < 1 || M + 1 + nv > PG_MN_MAXROWS_WIDE) FAIL(c, PGAS_E_ARG, "pgas_m_mniw_trisolve: M = %d outside [1, %d]", M, PG_MN_MAXROWS_WIDE - 1 - nv);
    if (n == 0) return PGAS_OK;
    DeviceGuard guard(c->device);
    if (M > PG_MN_MAXM || nv > 1 || c->mniw_valu == 2) {
        const size_t lds = (size_t)((M + 1 + nv) * (M + 2 + nv) / 2) * sizeof(double);
        if (lds > 64 * 1024) HIPCHK(c, hipFuncSetAttribute((const void*)k_mniw_trisolve_wide, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k_mniw_trisolve_wide, dim3((unsigned)n), dim3(64), lds, (hipStream_t)sh, n, M, nv, anc, Lfac, phi, m, cc);
        KCHK(c, "k_mniw_trisolve_wide");
        return PGAS_OK;
    }
    const int waves = 4;
    hipLaunchKernelGGL(k_mniw_trisolve, dim3((unsigned)((n + waves - 1) / waves)), dim3(64 * waves),
                       (size_t)waves * ((M + 2) * (M + 3) / 2) * sizeof(double), (hipStream_t)sh, n, M, anc, Lfac, phi, m, cc);
    KCHK(c, "k_mniw_trisolve");
    return PGAS_OK;
}

int pgas_m_expr_eval(pgas_ctx* c, int64_t n, const int32_t* code_dev, int32_t ninstr, const double* consts_dev, int32_t nconst, int32_t n_in, int32_t nreg,
                     const int32_t* out_regs, int32_t nout, const double* state_dev, int32_t nx, const int32_t* anc_dev, const double* input_dev, int32_t nu,
                     const double* const* iv_dev, const int32_t* iv_widths, int32_t n_iv, int32_t mode, const double* aux_dev, const double* mat_dev, double cR,
                     double* out_dev, void* sh) {
    if (!c) return PGAS_E_ARG;
    if (!state_dev || !out_dev || !out_regs || (ninstr > 0 && !code_dev) || (nconst > 0 && !consts_dev) || n < 0) FAIL(c, PGAS_E_ARG, "pgas_m_expr_eval: NULL argument");
    if (nreg < 1 || nreg > PG_EX_MAXREG || n_in + nconst > nreg) FAIL(c, PGAS_E_ARG, "pgas_m_expr_eval: %d registers outside [1, %d]", nreg, PG_EX_MAXREG);
    if (nout < 1 || nout > PG_EX_MAXOUT || n_iv < 0 || n_iv > PG_EX_MAXIV || mode < 0 || mode > 2) FAIL(c, PGAS_E_ARG, "pgas_m_expr_eval: bad shape argument");
    if ((nu > 0 && !input_dev) || (mode != 0 && (!aux_dev || !mat_dev))) FAIL(c, PGAS_E_ARG, "pgas_m_expr_eval: missing operand for this mode");
    int w = nx + nu;
    for (int i = 0; i < n_iv; ++i) w += iv_widths[i];
    if (w != n_in) FAIL(c, PGAS_E_ARG, "pgas_m_expr_eval: the program expects %d inputs, the operands have %d", n_in, w);
    for (int j = 0; j < nout; ++j)
        if (out_regs[j] < 0 || out_regs[j] >= nreg) FAIL(c, PGAS_E_ARG, "pgas_m_expr_eval: output register out of range");
    if (n == 0) return PGAS_OK;
    DeviceGuard guard(c->device);
    ExprArgs a{};
    a.state = state_dev; a.anc = anc_dev; a.u = input_dev; a.n_iv = n_iv; a.nx = nx; a.nu = nu;
    for (int i = 0; i < n_iv; ++i) { a.iv[i] = iv_dev[i]; a.ivw[i] = iv_widths[i]; }
    a.code = code_dev; a.consts = consts_dev; a.ninstr = ninstr; a.nconst = nconst; a.n_in = n_in;
    for (int j = 0; j < nout; ++j) a.out_reg[j] = out_regs[j];
    a.nout = nout; a.mode = mode; a.aux = aux_dev; a.mat = mat_dev; a.cR = cR; a.out = out_dev;
    hipLaunchKernelGGL(k_expr, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)sh, n, a);
    KCHK(c, "k_expr");
    return PGAS_OK;
}

int pgas_m_check(pgas_ctx* c, void* sh) {
    if (!c) return PGAS_E_ARG;
    if (!c->d_fail) return PGAS_OK;
    DeviceGuard guard(c->device);
    hipStream_t st = (hipStream_t)sh;
    int32_t bad = 0;
    HIPCHK(c, hipMemcpyAsync(&bad, c->d_fail, sizeof bad, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipMemsetAsync(c->d_fail, 0, sizeof(int32_t), st));
    HIPCHK(c, hipStreamSynchronize(st));
    if (bad) FAIL(c, PGAS_E_STATE, "pgas_m_mniw_solve: %d matrices eta1 were not positive definite since the last check", bad);
    return PGAS_OK;
}

int pgas_m_stats_gather_update(pgas_ctx* c, int64_t n, int32_t M, double scale, const int32_t* anc, const double* T0i, const double* T1i,
                               const double* T2i, const double* T3i, const double* phi, const double* xi, double* T0o, double* T1o, double* T2o,
                               double* T3o, void* sh) {
    return pgas_m_stats_gather_update_n(c, n, M, 1, scale, anc, T0i, T1i, T2i, T3i, phi, xi, T0o, T1o, T2o, T3o, sh);
}

int pgas_m_stats_gather_update_n(pgas_ctx* c, int64_t n, int32_t M, int32_t nv, double scale, const int32_t* anc, const double* T0i, const double* T1i,
                                 const double* T2i, const double* T3i, const double* phi, const double* xi, double* T0o, double* T1o, double* T2o,
                                 double* T3o, void* sh) {
    if (!c) return PGAS_E_ARG;
    if (nv < 1 || nv > 8) FAIL(c, PGAS_E_ARG, "pgas_m_stats_gather_update: %d components of the interface variable outside [1, 8]", nv);
    if (!T0i || !T1i || !T2i || !T3i || !phi || !xi || !T0o || !T1o || !T2o || !T3o || n < 0) FAIL(c, PGAS_E_ARG, "pgas_m_stats_gather_update: NULL argument");
    if (T0i == T0o || T1i == T1o || T2i == T2o || T3i == T3o) FAIL(c, PGAS_E_ARG, "pgas_m_stats_gather_update: input and output alias");
    if (M < 1 || M > PG_MN_MAXM_WIDE) FAIL(c, PGAS_E_ARG, "pgas_m_stats_gather_update: M = %d outside [1, %d]", M, PG_MN_MAXM_WIDE);
    if (n == 0) return PGAS_OK;
    DeviceGuard guard(c->device);
    hipLaunchKernelGGL(k_stats_gather_update, dim3((unsigned)n), dim3(256), 0, (hipStream_t)sh, n, M, nv, scale, anc, T0i, T1i, T2i, T3i, phi, xi, T0o, T1o, T2o, T3o);
    KCHK(c, "k_stats_gather_update");
    return PGAS_OK;
}

int pgas_m_weighted_stats(pgas_ctx* c, int64_t n, int32_t M, const double* w, const double* T0, const double* T1, const double* T2, const double* T3,
                          double* S0, double* S1, double* S2, double* S3, void* sh) {
    return pgas_m_weighted_stats_n(c, n, M, 1, w, T0, T1, T2, T3, S0, S1, S2, S3, sh);
}

int pgas_m_weighted_stats_n(pgas_ctx* c, int64_t n, int32_t M, int32_t nv, const double* w, const double* T0, const double* T1, const double* T2, const double* T3,
                            double* S0, double* S1, double* S2, double* S3, void* sh) {
    if (!c) return PGAS_E_ARG;
    if (nv < 1 || nv > 8) FAIL(c, PGAS_E_ARG, "pgas_m_weighted_stats: %d components of the interface variable outside [1, 8]", nv);
    if (!w || !T0 || !T1 || !T2 || !T3 || !S0 || !S1 || !S2 || !S3 || n < 1) FAIL(c, PGAS_E_ARG, "pgas_m_weighted_stats: bad argument");
    if (M < 1 || M > PG_MN_MAXM_WIDE) FAIL(c, PGAS_E_ARG, "pgas_m_weighted_stats: M = %d outside [1, %d]", M, PG_MN_MAXM_WIDE);
    DeviceGuard guard(c->device);
    hipStream_t st = (hipStream_t)sh;
    const int ncol = M * M + M * nv + nv * nv + 1;
    const int64_t nchunk = (n + PG_WS_CHUNK - 1) / PG_WS_CHUNK;
    if (nchunk > 65535) FAIL(c, PGAS_E_ARG, "pgas_m_weighted_stats: n = %lld too large", (long long)n);
    const size_t need = (size_t)nchunk * ncol * sizeof(double);
    if (c->ws_bytes < need) {
        HIPCHK(c, hipStreamSynchronize(st));
        hipFree(c->ws_partial);
        c->ws_partial = nullptr;
        c->ws_bytes = 0;
        HIPCHK(c, hipMalloc(&c->ws_partial, need));
        c->ws_bytes = need;
    }
    hipLaunchKernelGGL(k_weighted_stats_partial, dim3((ncol + 255) / 256, (unsigned)nchunk), dim3(256), 0, st, n, M, nv, w, T0, T1, T2, T3, c->ws_partial);
    KCHK(c, "k_weighted_stats_partial");
    hipLaunchKernelGGL(k_weighted_stats_final, dim3((ncol + 255) / 256), dim3(256), 0, st, (int)nchunk, M, nv, c->ws_partial, S0, S1, S2, S3);
    KCHK(c, "k_weighted_stats_final");
    return PGAS_OK;
}
