// pgas_resample.hip.h -- the weight recursion of the conditional-SMC sweep on gfx950: hierarchical CDF
// (DESIGN.md section 4.4), systematic-resampling search (src/Filtering.py:28-35), ancestor draw of the
// conditioned particle (src/PGAS.py:121-127), weight update (src/PGAS.py:137-147).
//
// Kernels:
//   k_groups      one wave per (CDF, group of 64 segments): per-segment records (e, sc, m) and the group record (KG, TG)
//   k_step<LOCAL> one launch per time step of the sweep: search of step t-1, weight update, softmax scans of step t,
//                 ancestor workgroup.  LOCAL = every workgroup scans all groups itself from the raw segment partials
//                 (single device, <= 1024 segments: no other launch on the critical path); otherwise the records come
//                 from k_groups (any size, any number of ranks: a workgroup reads the <= 128 group records and the
//                 segments of the one or two groups its slots fall into)
//   k_count       #{k : num_k < u S} against a stored cumsum: final index (src/PGAS.py:224-225), reference ancestor of pgas_step
//   k_back, k_back_corrected, k_systematic   the step API / src/Filtering.py entry points on the same search
#pragma once

#include "pgas_kernels.hip.h"

#define PG_GRP PGAS_GRP
#define PG_MAX_GRP (PG_MAX_NSEG / PG_GRP)
#define PG_WIN_SEG 1024                      // segments a workgroup keeps in LDS (its search window)
#define PG_WIN_GRP (PG_WIN_SEG / PG_GRP)
#define PG_LOCAL_NSEG PG_WIN_SEG             // k_step<true>: the window is the whole device
#ifndef PG_FSTAGE
#define PG_FSTAGE 2                          // source segments staged at once (LDS budget: five workgroups per CU)
#endif
#define PG_NCAND 8                           // staged candidates per workgroup before falling back to per-slot bisection
#define PG_DEXP_ZERO (-32768)
#ifdef PG_COLD_NOINLINE
#define PG_COLD_ATTR __noinline__
#else
#define PG_COLD_ATTR __forceinline__
#endif
static_assert(PG_MAX_GRP <= 128, "top_scan_wave handles two blocks of 64 groups");
#ifdef PG_DEBUG_DUMP
__device__ double g_dbg[4][64];
#endif

// (rank, cdf, local segment) -> word of the gathered partial arrays
__device__ __forceinline__ size_t partial_at(const ScanBufs& sb, int cdf, int b) {
    return (size_t)(b / sb.nseg_l) * sb.rank_stride + (size_t)cdf * sb.nsegp + (size_t)(b % sb.nseg_l);
}

// Group level (one wave = one group, lane = segment): kk = kref of the lane's segment (-inf when it does not exist), ss its total.
__device__ __forceinline__ void group_scan_wave(double kk, uint64_t ss, double& e, double& sc, double& m, double& Kg) {
    const int lane = threadIdx.x & 63;
    Kg = wave_max(kk);
    sc = pgas_lvl_scale(kk, Kg);
    const double t = sc * (pgas_u64_to_double(ss) * PGAS_FIX_INV);
    const double incl = wave_scan_add(t);
    const double up = __shfl_up(incl, 1);
    e = lane ? up : 0.0;
    m = wave_scan_max(e + t);
}

// Top level by one wave: lane l owns groups l and 64 + l (n1 <= 128).  Kg = -inf, Tg = 0 for groups that do not exist.
__device__ __forceinline__ double top_scan_wave(int n1, const double (&Kg)[2], const double (&Tg)[2], double (&E)[2], double (&sig)[2],
                                                double (&CM)[2]) {
    const int lane = threadIdx.x & 63;
    const double K = wave_max(__builtin_fmax(Kg[0], Kg[1]));
    double TT[2], inc[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        sig[h] = pgas_lvl_scale(Kg[h], K);
        TT[h] = sig[h] * Tg[h];
    }
    inc[0] = wave_scan_add(TT[0]);
    inc[1] = n1 > 64 ? wave_scan_add(TT[1]) : 0.0;
    const double BT0 = readlane_f64(inc[0], 63);
    double W[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const double up = __shfl_up(inc[h], 1);
        E[h] = (h ? BT0 : 0.0) + (lane ? up : 0.0);
        W[h] = (lane + 64 * h < n1) ? E[h] + TT[h] : 0.0;
    }
    CM[0] = wave_scan_max(W[0]);
    const double M0 = readlane_f64(CM[0], 63);
    CM[1] = M0;
    double S = M0;
    if (n1 > 64) {
        CM[1] = __builtin_fmax(wave_scan_max(W[1]), M0);
        S = readlane_f64(CM[1], 63);
    }
    return S;
}

__device__ __forceinline__ int dexp_of(double sc) { return sc > 0.0 ? (int)((pgas_d2bits(sc) >> 52) & 0x7ff) - 1023 : PG_DEXP_ZERO; }
__device__ __forceinline__ double sc_of(int dexp) { return dexp == PG_DEXP_ZERO ? 0.0 : ldexp(1.0, dexp); }

// CDF numerator of one particle: c = its integer cumsum inside the segment
struct SegParams {
    double e, sc, mp;     // segment: exclusive prefix inside the group, scale, running maximum before it
    double E, sg, cp;     // group: exclusive prefix, scale, running maximum before it
};
__device__ __forceinline__ double num_of(uint64_t c, const SegParams& p) {
    const double v = __builtin_fmax(p.mp, p.e + p.sc * (pgas_u64_to_double(c) * PGAS_FIX_INV));
    return __builtin_fmax(p.cp, p.E + p.sg * v);
}

// ------------------------------------------------------------------------------------------
// k_groups
// ------------------------------------------------------------------------------------------
// group scan of (cdf, g) by the calling wave: per-segment records and the group record
__device__ __forceinline__ void group_records_wave(const ScanBufs& sb, int nseg, int cdf, int g, bool coherent) {
    const int lane = threadIdx.x & 63;
    const int b = g * PG_GRP + lane;
    double kk = -__builtin_inf();
    uint64_t ss = 0;
    if (b < nseg) {
        const size_t at = partial_at(sb, cdf, b);
        if (coherent) {   // written by other workgroups of this launch: agent-scope loads behind the caller's acquire
            kk = __hip_atomic_load(&sb.segk[at], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ss = __hip_atomic_load(&sb.segs[at], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            kk = sb.segk[at];
            ss = sb.segs[at];
        }
    }
    double e, sc, m, Kg;
    group_scan_wave(kk, ss, e, sc, m, Kg);
    const int nb = nseg - g * PG_GRP < PG_GRP ? nseg - g * PG_GRP : PG_GRP;
    const double Tg = readlane_f64(m, nb - 1);
    if (b < nseg) {
        const size_t o = (size_t)cdf * sb.nsegp_g + b;
        sb.tab_e[o] = e;
        sb.tab_sc[o] = sc;
        sb.tab_m[o] = m;
    }
    if (lane == 0) {
        sb.grp_K[cdf * PG_MAX_GRP + g] = Kg;
        sb.grp_T[cdf * PG_MAX_GRP + g] = Tg;
    }
}

__global__ __launch_bounds__(256) void k_groups(int nseg, int ncdf, ScanBufs sb) {
    // 32 short waves on the sweep's critical path compete with k_propagate's long-running ones for issue slots: raised wave
    // priority takes the launch from 6.7 to 5.3 us under load (81.5 -> 79.9 ms per sweep); the same in k_step changes nothing
    __builtin_amdgcn_s_setprio(3);
    const int n1 = (nseg + PG_GRP - 1) / PG_GRP;
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= n1 * ncdf) return;
    group_records_wave(sb, nseg, w / n1, w % n1, false);
}

// ------------------------------------------------------------------------------------------
// Search window of a workgroup (256 threads) in LDS
// ------------------------------------------------------------------------------------------
// how the window of a workgroup is filled
#define PG_WM_GROUPS 0   // group records of k_groups + a top scan by one wave: any size, any number of ranks
#define PG_WM_LOCAL 1    // every workgroup scans all groups itself from the raw partials (single device, <= 1024 segments)
template <int LOCAL>
struct WinSmemT {
    double cm[PG_WIN_SEG];   // running maximum of the CDF at the end of every window segment, +inf beyond the window
    union {
        double num[PG_FSTAGE][PGAS_SEG];   // staged numerators
        struct {
            double e[PG_WIN_SEG], mp[PG_WIN_SEG];
        } tab;                              // per-segment records of the window (valid until the first staging)
        ScanSmem scan;
        int a[PGAS_SEG];   // ancestors, slot-major -> particle-major exchange
    } u;
    short dexp[PG_WIN_SEG];   // log2 of the segment scales
    double gE[PG_WIN_GRP], gS[PG_WIN_GRP], gCP[PG_WIN_GRP];   // window groups: E, sigma, CM of the group before
    double topCM[LOCAL ? 1 : PG_MAX_GRP], topE[LOCAL ? 1 : PG_MAX_GRP], topS[LOCAL ? 1 : PG_MAX_GRP];   // PG_WM_GROUPS only
    double grK[PG_WIN_GRP], grT[PG_WIN_GRP];   // LOCAL: group records found by the waves
    int cand_b[PG_NCAND];
    double cand_cy[PG_NCAND];
    SegParams cand_p[PG_NCAND];
    double S;
    int g_lo, g_hi;
    int cnt[2];
    unsigned long long wsum[PG_BLK / 64];
};

// Fills the window of CDF `cdf` for thresholds tau in [U_first S, U_last S].
// LOCAL: the window is the whole device (nseg <= 1024): every wave scans four groups from the raw partials.
// otherwise: the group records of k_groups give the top level; the window holds the groups [g_lo, g_hi] (at most 16).
// Out: S, first global segment of the window, number of window segments; returns false when the thresholds span more than the
// window can hold (the caller then searches slot by slot through global memory).
template <int LOCAL>
__device__ __forceinline__ bool window_head(WinSmemT<LOCAL>& sm, const ScanBufs& sb, int cdf, int nseg, double U_first, double U_last,
                                            double& S, int& win_b0, int& nwin) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n1 = (nseg + PG_GRP - 1) / PG_GRP;
    if constexpr (LOCAL == PG_WM_LOCAL) {
        double mreg[PG_WIN_GRP / 4];
#pragma unroll
        for (int e4 = 0; e4 < PG_WIN_GRP / 4; ++e4) {
            const int g = wave + 4 * e4, b = (g << 6) + lane;
            double kk = -__builtin_inf();
            uint64_t ss = 0;
            if (b < nseg) {
                const size_t at = partial_at(sb, cdf, b);
                kk = sb.segk[at];
                ss = sb.segs[at];
            }
            double e, sc, m, Kg;
            group_scan_wave(kk, ss, e, sc, m, Kg);
            const double upm = __shfl_up(m, 1);
            sm.u.tab.e[b] = e;
            sm.u.tab.mp[b] = lane ? upm : 0.0;
            sm.dexp[b] = (short)dexp_of(sc);
            mreg[e4] = m;
            int nb = nseg - g * PG_GRP;
            nb = nb < 1 ? 1 : (nb > PG_GRP ? PG_GRP : nb);
            const double Tg = readlane_f64(m, nb - 1);
            if (lane == 0) {
                sm.grK[g] = Kg;
                sm.grT[g] = Tg;
            }
        }
        __syncthreads();
        const double Kg2[2] = {lane < n1 ? sm.grK[lane & (PG_WIN_GRP - 1)] : -__builtin_inf(), -__builtin_inf()};
        const double Tg2[2] = {lane < n1 ? sm.grT[lane & (PG_WIN_GRP - 1)] : 0.0, 0.0};
        double E[2], sig[2], CM[2];
        S = top_scan_wave(n1, Kg2, Tg2, E, sig, CM);
#pragma unroll
        for (int e4 = 0; e4 < PG_WIN_GRP / 4; ++e4) {
            const int g = wave + 4 * e4, b = (g << 6) + lane;
            const double Eg = readlane_f64(E[0], g), sg = readlane_f64(sig[0], g), cpr = readlane_f64(CM[0], g ? g - 1 : 0);
            const double cp = g ? cpr : 0.0;
            sm.cm[b] = b < nseg ? __builtin_fmax(cp, Eg + sg * mreg[e4]) : __builtin_inf();
        }
        if (wave == 0) {
            const double upc = __shfl_up(CM[0], 1);
            if (lane < PG_WIN_GRP) {
                sm.gE[lane] = E[0];
                sm.gS[lane] = sig[0];
                sm.gCP[lane] = lane ? upc : 0.0;
            }
        }
        __syncthreads();
        win_b0 = 0;
        nwin = nseg;
        (void)U_first;
        (void)U_last;
        return true;
    } else {
        if (wave == 0) {
            double Kg2[2], Tg2[2], E[2], sig[2], CM[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int g = lane + 64 * h;
                Kg2[h] = g < n1 ? sb.grp_K[cdf * PG_MAX_GRP + g] : -__builtin_inf();
                Tg2[h] = g < n1 ? sb.grp_T[cdf * PG_MAX_GRP + g] : 0.0;
            }
            const double Sv = top_scan_wave(n1, Kg2, Tg2, E, sig, CM);
            const double tf = U_first * Sv, tl = U_last * Sv;
            int clo = 0, chi = 0;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int g = lane + 64 * h;
                if (g < n1) {
                    sm.topCM[g] = CM[h];
                    sm.topE[g] = E[h];
                    sm.topS[g] = sig[h];
                }
                clo += __popcll(__ballot(g < n1 && CM[h] < tf));
                chi += __popcll(__ballot(g < n1 && CM[h] < tl));
            }
            if (lane == 0) {
                sm.S = Sv;
                sm.g_lo = clo;
                sm.g_hi = chi > n1 - 1 ? n1 - 1 : chi;
            }
            PG_STAMP(9);
        }
        __syncthreads();
        PG_STAMP(10);
        S = sm.S;
        const int g_lo = sm.g_lo, g_hi = sm.g_hi;
        int ngw = g_hi - g_lo + 1;   // <= 0 when every threshold lies beyond the total (cannot happen for U < 1, kept safe)
        const bool covered = ngw <= PG_WIN_GRP;
        ngw = ngw < 0 ? 0 : (ngw > PG_WIN_GRP ? PG_WIN_GRP : ngw);
        win_b0 = g_lo * PG_GRP;
        nwin = ngw * PG_GRP < nseg - win_b0 ? ngw * PG_GRP : nseg - win_b0;
        if (nwin < 0) nwin = 0;
        const size_t o = (size_t)cdf * sb.nsegp_g;
        // The records of the window's groups, one row of loads per four groups (wave w takes groups w, w + 4, ...), clamped indices and
        // no guard around a load: guarded loads make the compiler wait for each in turn.  A window is one or two groups almost always,
        // so the usual case issues ONE row (three loads per lane); loading all sixteen groups regardless costs 24 KB of L1 traffic per
        // workgroup, 0.7 us when the four workgroups of a CU do it at once.
        constexpr int NR = PG_WIN_SEG / PG_BLK;
        auto fill = [&](int i, double e, double sc, double m) {
            const int gw = wave + 4 * i, wb = (gw << 6) + lane, g = g_lo + gw, b = win_b0 + wb;
            double cmv = __builtin_inf();
            if (gw < ngw) {   // wave-uniform
                if (b >= nseg) e = 0.0, sc = 0.0, m = 0.0;
                const double upm = __shfl_up(m, 1);
                const double Eg = sm.topE[g], sg = sm.topS[g], cp = g ? sm.topCM[g - 1] : 0.0;
                if (b < nseg) cmv = __builtin_fmax(cp, Eg + sg * m);
                sm.u.tab.e[wb] = e;
                sm.u.tab.mp[wb] = lane ? upm : 0.0;
                sm.dexp[wb] = (short)dexp_of(sc);
                if (lane == 0) {
                    sm.gE[gw] = Eg;
                    sm.gS[gw] = sg;
                    sm.gCP[gw] = cp;
                }
            }
            sm.cm[wb] = cmv;
        };
        auto at_of = [&](int i) {
            const int b = win_b0 + ((wave + 4 * i) << 6) + lane;
            return o + (size_t)(b < nseg ? b : nseg - 1);
        };
        if (ngw <= 4) {   // uniform
            const size_t at = at_of(0);
            const double e = sb.tab_e[at], sc = sb.tab_sc[at], m = sb.tab_m[at];
            PG_STAMP(11);
            fill(0, e, sc, m);
#pragma unroll
            for (int i = 1; i < NR; ++i) sm.cm[((wave + 4 * i) << 6) + lane] = __builtin_inf();
        } else {
            double ve[NR], vsc[NR], vm[NR];
#pragma unroll
            for (int i = 0; i < NR; ++i) {
                const size_t at = at_of(i);
                ve[i] = sb.tab_e[at];
                vsc[i] = sb.tab_sc[at];
                vm[i] = sb.tab_m[at];
            }
#pragma unroll
            for (int i = 0; i < NR; ++i) fill(i, ve[i], vsc[i], vm[i]);
        }
        __syncthreads();
        return covered;
    }
}

// #{wb : cm[wb] < tau} over the +inf padded window, by every wave for itself: one ballot over the group ends, one over the
// segments of the group found (two dependent LDS reads instead of ten)
template <class SM>
__device__ __forceinline__ int win_lower_bound(const SM& sm, double tau) {
    // branch-free on purpose: a build with an early `return PG_WIN_SEG` between the two ballots produced wrong ancestors in k_back /
    // k_step for some workgroups although each part was right on its own (and a barrier anywhere nearby hid it); see DESIGN.md 8
    const int lane = threadIdx.x & 63;
    const double ge = sm.cm[((lane & (PG_WIN_GRP - 1)) << 6) + 63];
    const int gw = __popcll(__ballot(lane < PG_WIN_GRP && ge < tau));
    const int g0 = __builtin_amdgcn_readfirstlane(gw < PG_WIN_GRP ? gw : PG_WIN_GRP - 1) << 6;
    const int inner = __popcll(__ballot(sm.cm[g0 + lane] < tau));
    return gw >= PG_WIN_GRP ? PG_WIN_SEG : g0 + inner;
}

// the same count for a threshold that differs from lane to lane (degenerate path): branch-free bisection
template <class SM>
__device__ __forceinline__ int win_lower_bound_lane(const SM& sm, double tau) {
    int p = 0;
#pragma unroll
    for (int step = PG_WIN_SEG / 2; step >= 1; step >>= 1)
        if (sm.cm[p + step - 1] < tau) p += step;
    return p;
}

template <class SM>
__device__ __forceinline__ SegParams win_params(const SM& sm, int wb) {
    SegParams p;
    p.e = sm.u.tab.e[wb];
    p.sc = sc_of(sm.dexp[wb]);
    p.mp = sm.u.tab.mp[wb];
    p.E = sm.gE[wb >> 6];
    p.sg = sm.gS[wb >> 6];
    p.cp = sm.gCP[wb >> 6];
    return p;
}

__device__ __forceinline__ int seg_count(int N, int b) {
    const int64_t base = (int64_t)b * PGAS_SEG;
    return (N - base) < PGAS_SEG ? (int)(N - base) : PGAS_SEG;
}

// c1 data of GLOBAL source segment bs (this device's buffer, or a peer's through its xGMI mapping)
__device__ __forceinline__ const uint64_t* c1_segment(const ScanBufs& sb, const Peers& pr, int bs) {
    return pr.world > 1 ? pr.c1[bs / pr.nseg_l] + (size_t)(bs % pr.nseg_l) * PGAS_SEG : sb.c1 + (size_t)bs * PGAS_SEG;
}

// first k of segment b with num_k >= tau (n when there is none)
__device__ __forceinline__ int seg_lower_bound(const uint64_t* __restrict__ c, int n, const SegParams& p, double tau) {
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (num_of(c[mid], p) < tau) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// one slot through the global tables (thresholds outside the window: heavily degenerate weights only)
template <class SM>
__device__ __forceinline__ int slot_search_global(const SM& sm, const ScanBufs& sb, const Peers& pr, int cdf, int nseg, int N, double tau) {
    const int n1 = (nseg + PG_GRP - 1) / PG_GRP;
    int lo = 0, hi = n1;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (sm.topCM[mid] < tau) lo = mid + 1; else hi = mid;
    }
    if (lo >= n1) return N - 1;
    const int g = lo;
    SegParams p;
    p.E = sm.topE[g];
    p.sg = sm.topS[g];
    p.cp = g ? sm.topCM[g - 1] : 0.0;
    const size_t o = (size_t)cdf * sb.nsegp_g + (size_t)g * PG_GRP;
    const int nb = nseg - g * PG_GRP < PG_GRP ? nseg - g * PG_GRP : PG_GRP;
    lo = 0, hi = nb - 1;   // the group's last segment reaches W_g >= tau
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (__builtin_fmax(p.cp, p.E + p.sg * sb.tab_m[o + mid]) < tau) lo = mid + 1; else hi = mid;
    }
    const int b = g * PG_GRP + lo;
    p.e = sb.tab_e[o + lo];
    p.sc = sb.tab_sc[o + lo];
    p.mp = lo ? sb.tab_m[o + lo - 1] : 0.0;
    const int64_t ai = (int64_t)b * PGAS_SEG + seg_lower_bound(c1_segment(sb, pr, b), seg_count(N, b), p, tau);
    return ai > N - 1 ? N - 1 : (int)ai;
}

// Resampling slot j of thread tid inside its workgroup's 1024 slots: lanes of a wave take consecutive slots (for every j),
// so that the lower-bound probes of a wave fall on consecutive LDS words (no bank conflicts).
__device__ __forceinline__ int slot_of(int tid, int j) { return ((tid >> 6) << 8) + (j << 6) + (tid & 63); }

__device__ __forceinline__ double slot_U(double u1, int64_t i, int N, double invN, bool pow2) {
    const double x = u1 + (double)i;
    return pow2 ? x * invN : x / (double)N;  // exact either way when N is a power of two
}

// ------------------------------------------------------------------------------------------
// Systematic-resampling search (src/Filtering.py:28-35) for the 1024 slots of local segment `seg`: a[j] = ancestor of
// slot slot_of(tid, j), a GLOBAL particle index.  md.p0 / md.Ng / md.nseg_g place the device's shard in the global
// particle range (single device: 0 / N / nseg).  Leaves LDS free for reuse after a trailing barrier of the caller.
// ------------------------------------------------------------------------------------------
struct NoPrefetch {
    __device__ __forceinline__ void operator()() const {}
};
// `prefetch` is called once, right after the window's loads have come back: the place for loads the caller needs only after the
// search (issued any earlier they sit in front of the window's loads in the in-order vmcnt queue and delay them by an HBM miss).
template <int LOCAL, class PF = NoPrefetch>
__device__ __forceinline__ void resample_search(const DevModel& md, WinSmemT<LOCAL>& sm, double u1, const ScanBufs& sb, const Peers& pr,
                                                int seg, int (&a)[PG_PPT], PF prefetch = PF()) {
    const int tid = threadIdx.x;
    const int nseg = md.nseg_g, N = md.Ng;
    const bool pow2 = (N & (N - 1)) == 0;
    const double invN = 1.0 / (double)N;
    const int64_t loc_i = (int64_t)seg * PGAS_SEG;
    const int64_t base_i = md.p0 + loc_i;
    const int nslots = (md.N - loc_i) < PGAS_SEG ? (int)(md.N - loc_i) : PGAS_SEG;
    const double U_first = slot_U(u1, base_i, N, invN, pow2), U_last = slot_U(u1, base_i + nslots - 1, N, invN, pow2);
    double S;
    int win_b0, nwin;
    const bool covered = window_head<LOCAL>(sm, sb, 0, nseg, U_first, U_last, S, win_b0, nwin);
    prefetch();
    PG_STAMP(1);
    const bool valid = (S > 0.0) && (S < __builtin_inf());
    double tau[PG_PPT];
#pragma unroll
    for (int j = 0; j < PG_PPT; ++j) {
        const int64_t i = base_i + slot_of(tid, j);
        tau[j] = slot_U(u1, i, N, invN, pow2) * S;
        a[j] = valid ? N - 1 : (int)(i < N ? i : N - 1);
    }
    int ns = 0;   // non-empty source segments of this workgroup's slots, counted up to PG_NCAND + 1
    if (valid && covered) {   // uniform
        const double tau_first = U_first * S, tau_last = U_last * S;
        int b_lo = win_lower_bound(sm, tau_first), b_hi = win_lower_bound(sm, tau_last);
        if (b_hi > nwin - 1) b_hi = nwin - 1;   // slots beyond the last segment keep a = N-1
        // enumerate the non-empty source segments of [b_lo, b_hi] (a segment whose running max did not move owns no slot):
        // per window group one ballot of "the running maximum moved here"; the lane that owns a candidate writes its record.
        // Bounded work however long the run of empty segments between two heavy particles is.
        const int lane = tid & 63, wave = tid >> 6;
        for (int gw = b_lo >> 6; gw <= (b_hi >> 6) && ns <= PG_NCAND; ++gw) {   // uniform
            const int wb = (gw << 6) + lane;
            const double v = sm.cm[wb];
            const double vp = wb ? sm.cm[wb > 0 ? wb - 1 : 0] : (win_b0 ? sm.gCP[0] : 0.0);
            const bool cand = wb >= b_lo && wb <= b_hi && v > vp;
            const unsigned long long mask = __ballot(cand);
            const int rank = ns + __popcll(mask & ((1ull << lane) - 1ull));
            if (wave == 0 && cand && rank < PG_NCAND) {
                sm.cand_b[rank] = win_b0 + wb;
                sm.cand_cy[rank] = vp;
                sm.cand_p[rank] = win_params(sm, wb);
            }
            ns += __popcll(mask);
        }
        if (ns > PG_NCAND + 1) ns = PG_NCAND + 1;
#ifdef PG_DEBUG_DUMP
        if (seg == PG_DEBUG_DUMP && lane == 0) {
            double* d = g_dbg[wave];
            d[0] = S; d[1] = b_lo; d[2] = b_hi; d[3] = ns; d[4] = tau_first; d[5] = tau_last; d[6] = win_b0; d[7] = nwin;
            for (int k = 0; k < 24; ++k) d[8 + k] = sm.cm[k];
        }
#endif
    }
    PG_STAMP(2);
    if (valid && covered && ns > 0 && ns <= PG_NCAND) {
        // ---- common case: stage the numerators of PG_FSTAGE source segments at a time, back to back; they are
        // non-decreasing across the window (running-max carry), so one branch-free lower bound per slot settles
        // every slot that falls into the window
        __syncthreads();
        for (int k0w = 0; k0w < ns; k0w += PG_FSTAGE) {  // uniform
            const int ng = ns - k0w < PG_FSTAGE ? ns - k0w : PG_FSTAGE;
            int sb_idx[PG_FSTAGE] = {};
            SegParams sp[PG_FSTAGE];
#pragma unroll
            for (int g = 0; g < PG_FSTAGE; ++g)
                if (g < ng) {
                    sb_idx[g] = sm.cand_b[k0w + g];
                    sp[g] = sm.cand_p[k0w + g];
                }
            const double g_carry = sm.cand_cy[k0w];
            // global loads first, then the barrier that frees the table / the previous window
            // thread -> elements j * 256 + tid of every staged segment: 8-byte accesses with consecutive lanes on consecutive words, for the
            // global loads (512 contiguous bytes per wave instruction) and for the LDS stores (no bank conflicts; 32-byte-strided
            // 16-byte stores of four consecutive elements per lane put two lanes on every bank)
            uint64_t cq[PG_FSTAGE][PG_PPT];
#pragma unroll
            for (int g = 0; g < PG_FSTAGE; ++g)
                if (g < ng) {
                    const uint64_t* src = c1_segment(sb, pr, sb_idx[g]);
#pragma unroll
                    for (int j = 0; j < PG_PPT; ++j) cq[g][j] = src[j * PG_BLK + tid];
                }
            __syncthreads();
#pragma unroll
            for (int g = 0; g < PG_FSTAGE; ++g) {
                const int n = g < ng ? seg_count(N, sb_idx[g]) : 0;
#pragma unroll
                for (int j = 0; j < PG_PPT; ++j) {
                    const int k = j * PG_BLK + tid;
                    sm.u.num[g][k] = k < n ? num_of(cq[g][j], sp[g]) : __builtin_inf();
                }
            }
            __syncthreads();
            PG_STAMP(3);
            const double* __restrict__ num = &sm.u.num[0][0];
            int pos[PG_PPT] = {0, 0, 0, 0};
            static_assert((PG_FSTAGE * PGAS_SEG & (PG_FSTAGE * PGAS_SEG - 1)) == 0, "power-of-two window: pos + step - 1 never leaves it");
#pragma unroll
            for (int step = PG_FSTAGE * PGAS_SEG / 2; step >= 1; step >>= 1) {
#pragma unroll
                for (int j = 0; j < PG_PPT; ++j) {  // loads are unconditional so the four chains advance in lock step
                    const double v = num[pos[j] + step - 1];
                    pos[j] = (v < tau[j]) ? pos[j] + step : pos[j];
                }
            }
#pragma unroll
            for (int j = 0; j < PG_PPT; ++j) {
                if (g_carry < tau[j] && pos[j] < ng * PGAS_SEG) {
                    const int g = pos[j] >> 10, off = pos[j] & (PGAS_SEG - 1);
                    int sbg = sb_idx[0];
#pragma unroll
                    for (int q = 1; q < PG_FSTAGE; ++q) sbg = g == q ? sb_idx[q] : sbg;
                    const int64_t ai = (int64_t)sbg * PGAS_SEG + off;
                    a[j] = ai > N - 1 ? N - 1 : (int)ai;
                }
            }
        }
    } else if (valid && covered && ns > PG_NCAND) {
        // degenerate weights: per-slot bisection, segment level in LDS, particle level in global memory
#pragma unroll
        for (int j = 0; j < PG_PPT; ++j) {
            const int wb = win_lower_bound_lane(sm, tau[j]);
            if (wb < nwin) {
                const SegParams p = win_params(sm, wb);
                const int b = win_b0 + wb;
                const int64_t ai = (int64_t)b * PGAS_SEG + seg_lower_bound(c1_segment(sb, pr, b), seg_count(N, b), p, tau[j]);
                a[j] = ai > N - 1 ? N - 1 : (int)ai;
            }
        }
    } else if (valid && !covered) {
        if constexpr (LOCAL == PG_WM_GROUPS) {
#pragma unroll
            for (int j = 0; j < PG_PPT; ++j) a[j] = slot_search_global(sm, sb, pr, 0, nseg, N, tau[j]);
        }
    }
}

// Search + the bookkeeping every caller needs: conditioned slot, slot-major -> particle-major through LDS, ancestor trace.
template <int LOCAL>
__device__ __forceinline__ void resample_slots(const DevModel& md, WinSmemT<LOCAL>& sm, double u1, const ScanBufs& sb, const Peers& pr, int seg,
                                               int32_t* __restrict__ anc_out, int (&anc_pm)[PG_PPT], int ref_idx /* < 0: none */) {
    const int tid = threadIdx.x;
    int a[PG_PPT];
    resample_search<LOCAL>(md, sm, u1, sb, pr, seg, a);
    const int64_t loc_i = (int64_t)seg * PGAS_SEG, base_i = md.p0 + loc_i;
#pragma unroll
    for (int j = 0; j < PG_PPT; ++j)
        if (ref_idx >= 0 && base_i + slot_of(tid, j) == md.Ng - 1) a[j] = ref_idx;  // src/PGAS.py:127
    __syncthreads();
#pragma unroll
    for (int j = 0; j < PG_PPT; ++j) {
        sm.u.a[slot_of(tid, j)] = a[j];
        if (loc_i + slot_of(tid, j) < md.N) anc_out[loc_i + slot_of(tid, j)] = a[j];
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < PG_PPT; ++r) anc_pm[r] = sm.u.a[r * PG_BLK + tid];  // ancestor of local particle loc_i + r*BLK + tid
}

// ------------------------------------------------------------------------------------------
// #{k : num_k < U S} of CDF `cdf` by one workgroup, the searchsorted of src/PGAS.py:122-124 / :225, min'ed with N-1.
//   stored:    the per-particle cumsum comes from a buffer (cbuf of this device, or cb_ranks[r] through peer mappings)
//   otherwise: it is REBUILT for the one segment that matters from what the sweep keeps in HBM (AncIn, below)
// ------------------------------------------------------------------------------------------
// The ancestor CDF of a step (src/PGAS.py:117-124) is read in ONE segment only: the one the reference particle's uniform falls
// into.  The sweep therefore never stores its per-particle cumsum; the workgroup that draws the ancestor rebuilds that
// segment -- lw2_i = (la_s[i] + logw_{s-1}[i]) + h_s[i] with logw_{s-1}[i] = ln_{s-1}[i] - la_{s-1}[a_{s-1}[i]] (0 for s = 1),
// the same expressions, the segment's stored reference k, the same fixed-point numerators and an exact integer cumsum --
// and counts against it.  Bit-identical to a stored version.
struct AncIn {          // row pointers of every rank (this device's own rows or xGMI peer mappings), filled in by the host per launch
    const double* la_s[PG_MAX_RANKS];    // row s of la: log p(y_s | aux_s)
    const double* h_s[PG_MAX_RANKS];     // row s of h:  log N(ref_s; aux_s, S)
    const double* ln_p[PG_MAX_RANKS];    // row s-1 of ln (has_prev)
    const double* la_p[PG_MAX_RANKS];    // row s-1 of la (has_prev)
    const int32_t* anc_p[PG_MAX_RANKS];  // ancestors of step s-1 = row s-2 of the ancestor trace (has_prev)
    int32_t has_prev;                    // 0 for s = 1: logw_0 = 0
};

template <int LOCAL>
__device__ __forceinline__ int cdf_count_wg(WinSmemT<LOCAL>& sm, const ScanBufs& sb, const Peers& pr, int cdf, int nseg, int N, double U,
                                            const uint64_t* __restrict__ cbuf, const uint64_t* const* cb_ranks, const AncIn* rebuild) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double S;
    int win_b0, nwin;
    window_head<LOCAL>(sm, sb, cdf, nseg, U, U, S, win_b0, nwin);   // one threshold: always covered
    if (!((S > 0.0) && (S < __builtin_inf()))) return N - 1;
    const double tau = U * S;
    const int wb = win_lower_bound(sm, tau);
    if (wb >= nwin) return N - 1;
    const SegParams p = win_params(sm, wb);
    const int b = win_b0 + wb;
    const int n = seg_count(N, b);
    const int64_t base = (int64_t)b * PGAS_SEG;
    if (tid == 0) sm.cnt[0] = 0;
    int k = 0;
    if (rebuild == nullptr) {
        const uint64_t* __restrict__ cseg = (cb_ranks && pr.world > 1) ? cb_ranks[b / pr.nseg_l] + (size_t)(b % pr.nseg_l) * PGAS_SEG : cbuf + base;
        __syncthreads();
        for (int i = tid; i < n; i += PG_BLK) k += (num_of(cseg[i], p) < tau) ? 1 : 0;
    } else {
        // owner rank of the segment and its local rows
        const int r = pr.world > 1 ? b / pr.nseg_l : 0;
        const int64_t lbase = base - (int64_t)r * pr.Nl;   // local particle index of the segment's first particle on rank r
        const double kref = sb.segk[partial_at(sb, cdf, b)];
        const double* __restrict__ la_s = rebuild->la_s[r];
        const double* __restrict__ h_s = rebuild->h_s[r];
        const double* __restrict__ ln_p = rebuild->ln_p[r];
        const int32_t* __restrict__ anc_p = rebuild->anc_p[r];
        double arg[PG_PPT], ev[PG_PPT];
        {
            // loads in rounds, every round issued for the four particles together (clamped indices instead of guards)
            int64_t li[PG_PPT];
            double las[PG_PPT], hs[PG_PPT], lnp[PG_PPT], lap[PG_PPT];
            int apv[PG_PPT];
#pragma unroll
            for (int j = 0; j < PG_PPT; ++j) {
                const int i = PG_PPT * tid + j;
                li[j] = lbase + (i < n ? i : n - 1);
                las[j] = la_s[li[j]];
                hs[j] = h_s[li[j]];
                lnp[j] = 0.0;
                lap[j] = 0.0;
            }
            if (rebuild->has_prev) {   // uniform
#pragma unroll
                for (int j = 0; j < PG_PPT; ++j) {
                    apv[j] = anc_p[li[j]];   // GLOBAL index of the ancestor at step s-1
                    lnp[j] = ln_p[li[j]];
                }
                const double* rowp[PG_PPT];
                int off[PG_PPT];
#pragma unroll
                for (int j = 0; j < PG_PPT; ++j) {
                    const int ra = pr.world > 1 ? apv[j] / pr.Nl : 0;
                    off[j] = apv[j] - ra * pr.Nl;
                    rowp[j] = rebuild->la_p[ra];
                }
#pragma unroll
                for (int j = 0; j < PG_PPT; ++j) lap[j] = rowp[j][off[j]];
            }
#pragma unroll
            for (int j = 0; j < PG_PPT; ++j) {
                const int i = PG_PPT * tid + j;
                const double logw = rebuild->has_prev ? lnp[j] - lap[j] : 0.0;
                const double l1 = las[j] + logw;
                arg[j] = pgas_seg_arg(i < n ? l1 + hs[j] : -__builtin_inf(), kref);
            }
        }
        pgas_exp_n(arg, ev, PG_PPT);
        uint64_t loc[PG_PPT], run = 0;
#pragma unroll
        for (int j = 0; j < PG_PPT; ++j) {
            run += (ev[j] > 0.0) ? pgas_double_to_u64(__builtin_rint(ev[j] * PGAS_FIX_SCALE)) : 0ull;
            loc[j] = run;
        }
        const uint64_t incl = wave_incl_scan_u64(run);
        if (lane == 63) sm.wsum[wave] = incl;
        __syncthreads();
        uint64_t off = incl - run;
#pragma unroll
        for (int v = 0; v < PG_BLK / 64; ++v)
            if (v < wave) off += sm.wsum[v];
#pragma unroll
        for (int j = 0; j < PG_PPT; ++j) k += (PG_PPT * tid + j < n && num_of(off + loc[j], p) < tau) ? 1 : 0;
    }
    k = wave_sum_i(k);
    if (lane == 0 && k) atomicAdd(&sm.cnt[0], k);
    __syncthreads();
    const int64_t res = base + sm.cnt[0];
    return res > N - 1 ? N - 1 : (int)res;
}

// k_count: one workgroup; what = 0: reference ancestor of pgas_step (CDF 1 against c2) -> hdr->ref_idx,
//                          what = 1: final index (CDF 0 against c1)                    -> hdr->final_idx
__global__ __launch_bounds__(PG_BLK) void k_count(int N, int nseg, ScanBufs sb, Peers pr, int what, double u_val, const SweepParams* __restrict__ sp) {
    __shared__ WinSmemT<PG_WM_GROUPS> sm;
    const int cdf = what == 0 ? 1 : 0;
    const double u = sp ? sp->u_final : u_val;   // sweeps: the final-index uniform of the running sweep, device-resident
    const int r = cdf_count_wg<PG_WM_GROUPS>(sm, sb, pr, cdf, nseg, N, u, cdf ? sb.c2 : sb.c1, cdf ? pr.c2 : pr.c1, nullptr);
    if (threadIdx.x == 0) {
        if (what == 0) sb.hdr->ref_idx = r; else sb.hdr->final_idx = r;
    }
}

// ------------------------------------------------------------------------------------------
// step API back half: systematic resampling search + weight update (src/PGAS.py:137-147)
// ------------------------------------------------------------------------------------------
template <int NX>
__global__ __launch_bounds__(PG_BLK) void k_back(DevModel md, int t, double u1, const double* __restrict__ x_cur, ScanBufs sb, Peers pr,
                                                  int32_t* __restrict__ anc_out, double* __restrict__ logw_out) {
    __shared__ WinSmemT<PG_WM_GROUPS> sm;
    const int seg = blockIdx.x, tid = threadIdx.x;
    double xv[PG_PPT][NX];
    load_particles<NX>(md, x_cur, seg, xv);
    int anc[PG_PPT];
    resample_slots<PG_WM_GROUPS>(md, sm, u1, sb, pr, seg, anc_out, anc, sb.hdr->ref_idx);
    const double* __restrict__ yt = md.y + (size_t)t * md.ny;
#pragma unroll
    for (int r = 0; r < PG_PPT; ++r) {
        const int64_t i = (int64_t)seg * PGAS_SEG + r * PG_BLK + tid;
        if (i < md.N) logw_out[i] = loglik<NX>(md, yt, xv[r]) - sb.laux[anc[r]];
    }
}

// k_back_corrected: the second half of a step in the CORRECTED mode (resample_before_propagate; quirk Q1 removed):
//   a = systematic resampling search,  x_new_i = aux[a_i] + L_S z_i  (conditioned particle = ref_t),
//   logw_new_i = log p(y_t | x_new_i) - l_aux[a_i].
// aux holds the transition means k_front stored; the noise z_i is the same Philox draw the default mode uses for
// particle i at time t, so the two modes differ only in which mean the noise is added to.
template <int NX>
__global__ __launch_bounds__(PG_BLK) void k_back_corrected(DevModel md, const TransParams* __restrict__ tpp, int t, uint64_t seed, double u1,
                                                            const double* __restrict__ aux, const double* __restrict__ ref_t, ScanBufs sb, Peers pr,
                                                            int32_t* __restrict__ anc_out, double* __restrict__ x_new,
                                                            double* __restrict__ logw_out) {
    const TransParams tp = *tpp;
    __shared__ WinSmemT<PG_WM_GROUPS> sm;
    const int seg = blockIdx.x, tid = threadIdx.x;
    int anc[PG_PPT];
    resample_slots<PG_WM_GROUPS>(md, sm, u1, sb, pr, seg, anc_out, anc, sb.hdr->ref_idx);
    const double* __restrict__ yt = md.y + (size_t)t * md.ny;
    double z0[PG_PPT], z1[PG_PPT];
    {
        pgas_u32x4 w[PG_PPT];
#pragma unroll
        for (int r = 0; r < PG_PPT; ++r) {
            const int64_t pi = (int64_t)seg * PGAS_SEG + r * PG_BLK + tid;
            w[r] = pgas_rng_block(seed, PGAS_STREAM_PROP, 0u, (uint32_t)t, (uint64_t)(md.p0 + pi));
        }
        pgas_normal_pair_n(w, z0, z1, PG_PPT);
    }
    double xn[PG_PPT][NX];
#pragma unroll
    for (int r = 0; r < PG_PPT; ++r) {
        const int64_t pi = (int64_t)seg * PGAS_SEG + r * PG_BLK + tid;
        const int src = pi < md.N ? anc[r] : 0;
        const double z[2] = {z0[r], z1[r]};
#pragma unroll
        for (int k = 0; k < NX; ++k) {
            double v = aux[(size_t)src * NX + k];
#pragma unroll
            for (int l = 0; l <= k; ++l) v = PGAS_FMA(tp.LS[k * NX + l], z[l], v);
            xn[r][k] = (md.p0 + pi == md.Ng - 1) ? ref_t[k] : v;
        }
        if (pi < md.N) logw_out[pi] = loglik<NX>(md, yt, xn[r]) - sb.laux[src];
    }
    store_particles<NX>(md, x_new, seg, xn);
}

// systematic_SISR (src/Filtering.py:6-37) on a weight vector whose segment scans (k_segscan) and group records (k_groups) are in sb
__global__ __launch_bounds__(PG_BLK) void k_systematic(DevModel md, double u, const double* __restrict__ u_dev, ScanBufs sb, Peers pr,
                                                      int32_t* __restrict__ idx_out) {
    __shared__ WinSmemT<PG_WM_GROUPS> sm;
    int anc[PG_PPT];
    resample_slots<PG_WM_GROUPS>(md, sm, u_dev ? u_dev[0] : u, sb, pr, blockIdx.x, idx_out, anc, -1);
}

// ------------------------------------------------------------------------------------------
// k_step: one launch per time step of the sweep (condSequentialMonteCarlo.__call__, src/PGAS.py:199-221).
// Launch t in [1, T]: resamples step t-1 (t > 1: search, ancestors, logw_{t-1} = ln_{t-1} - la_{t-1}[a]) and scans step t
// (t < T: both softmaxes, segment partials for the group scans of the next launch).
//
// Grid = nseg + 1 workgroups.  Workgroup 0 only draws the reference particle's ancestor (src/PGAS.py:121-127) and publishes
// it as one 8-byte {launch tag, index} word; workgroup b + 1 owns segment b.  The workgroup that owns the conditioned
// particle reads that word late (after its own search).  Workgroup 0 never waits for anyone, so the hand-off cannot
// deadlock whatever the dispatch order; if the word has not arrived within the spin budget the owner computes the
// ancestor itself (same code, same result).
// ------------------------------------------------------------------------------------------
#define PG_RS_SEARCH 1  // resample step t-1 (sb_prev valid) and form logw_{t-1}; otherwise logw_{t-1} = 0 (t = 1)
#define PG_RS_SCAN 2    // scan step t's weights into sb_next; otherwise only emit logw_{t-1} (after the last step)

struct StepArgs {
    int t, mode;
    const SweepParams* sp;   // epoch (hand-off tag) of the running sweep
    const double* u_res;     // (T + 1) resampling uniforms of the sweep, device (k_sweep_begin); launch t searches with u_res[t-1]
    const double* u_anc;     // (T + 1) ancestor uniforms; the ancestor workgroup of launch t draws with u_anc[t-1]
    const double* la_t;      // (np) row t of la_buf, this device
    const double* h_t;       // (np) row t of h_buf
    const double* ln_prev;   // (np) row t-1 of ln_buf
    AncIn anc_in;            // rows of step s = t-1 on every rank: the ancestor workgroup rebuilds its segment from them, and la_s is where
                             // every workgroup reads log p(y_{t-1} | aux_{t-1}) of its (possibly remote) ancestors
    int32_t* anc_out;        // (N) ancestors of step t-1
    double* logw_out;        // optional (N) logw_{t-1}
};

// Out-of-line copy for the path that practically never runs (the hand-off word of the ancestor workgroup did not arrive within
// the spin budget): keeps the second instance of the count out of k_step's instruction stream.
template <int LOCAL>
__device__ PG_COLD_ATTR int cdf_count_wg_cold(WinSmemT<LOCAL>& sm, const ScanBufs& sb, const Peers& pr, int nseg, int N, double U, const AncIn& in) {
    return cdf_count_wg<LOCAL>(sm, sb, pr, 1, nseg, N, U, nullptr, nullptr, &in);
}

// TAIL (single device, not LOCAL): the group scans of step t run inside this launch -- the workgroup that completes a group of
// 64 segments (arrival counter) scans it -- instead of a k_groups launch between two k_step launches.  Hand-off per
// cdna_hip_programming.md Guideline 16 (write-through payload, drained, then the counter; the last arriver acquires).
#ifndef PG_STEP_OCC
#define PG_STEP_OCC 4   // waves per SIMD k_step is compiled for: 4 -> 108 VGPRs and no scratch; 5 (the LDS limit, 5 x 31 KB) -> 96 VGPRs + 36 B/lane of scratch = 15 MB more HBM traffic per launch at the same speed
#endif
template <int LOCAL, bool TAIL = false>
__global__ __launch_bounds__(PG_BLK, PG_STEP_OCC) void k_step(DevModel md, StepArgs ar, ScanBufs sb_prev, ScanBufs sb_next, Peers pr) {
    __shared__ WinSmemT<LOCAL> sm;
    const int tid = threadIdx.x;
    const int N = md.N;
    // unique per launch and per sweep (replays of a captured sweep included): 2654435761 is odd, so two launches collide only if their
    // step numbers differ by a multiple of it
    const unsigned tag = ar.sp->epoch * 2654435761u + (unsigned)ar.t;
    const double u1_prev = (ar.mode & PG_RS_SEARCH) ? ar.u_res[ar.t - 1] : 0.0, u2_prev = (ar.mode & PG_RS_SEARCH) ? ar.u_anc[ar.t - 1] : 0.0;
#ifdef PG_STEP_PRIO
    __builtin_amdgcn_s_setprio(PG_STEP_PRIO);   // the weight recursion is the latency chain of the sweep: let its few vector instructions go first
#endif
    if (blockIdx.x == 0) {  // ---- ancestor workgroup
        if (!(ar.mode & PG_RS_SEARCH)) return;
        const int r = cdf_count_wg<LOCAL>(sm, sb_prev, pr, 1, md.nseg_g, md.Ng, u2_prev, nullptr, nullptr, &ar.anc_in);
        if (tid == 0)
            __hip_atomic_store(&sb_prev.hdr->ref_granule, ((unsigned long long)tag << 32) | (unsigned)r, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    const int seg = blockIdx.x - 1;
    const int64_t base_i = (int64_t)seg * PGAS_SEG;
    PG_STAMP(0);
    // ln_{t-1} of the own particles is loaded inside the search, behind the window's loads (unconditionally: the rows are padded to
    // whole segments).  Issued first thing and behind a `mode & SEARCH ?` guard, the compiler waited for every one of them at the join:
    // two HBM round trips before the window was even started.
    double lnv[PG_PPT] = {0.0, 0.0, 0.0, 0.0};
    double lwp[PG_PPT] = {0.0, 0.0, 0.0, 0.0};
    if (ar.mode & PG_RS_SEARCH) {
        int a[PG_PPT];
        resample_search<LOCAL>(md, sm, u1_prev, sb_prev, pr, seg, a, [&]() {
#pragma unroll
            for (int r = 0; r < PG_PPT; ++r) lnv[r] = ld_stream(&ar.ln_prev[(size_t)base_i + r * PG_BLK + tid]);
        });
        const bool last_wg = md.p0 + base_i + PGAS_SEG >= md.Ng;  // uniform: owns the conditioned particle
        if (last_wg) {
            // ancestor of the conditioned particle, published by workgroup 0 (src/PGAS.py:127)
            __syncthreads();
            if (tid == 0) {
                unsigned long long g = 0;
                int spins = 0;
                for (; spins < (1 << 16); ++spins) {
                    g = __hip_atomic_load(&sb_prev.hdr->ref_granule, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if ((unsigned)(g >> 32) == tag) break;
                    __builtin_amdgcn_s_sleep(8);
                }
                sm.cnt[1] = ((unsigned)(g >> 32) == tag) ? (int)(unsigned)g : -1;
            }
            __syncthreads();
            int ref_idx = sm.cnt[1];
            if (ref_idx < 0) {  // uniform: the word never arrived -- draw the ancestor here
                __syncthreads();
                ref_idx = cdf_count_wg_cold<LOCAL>(sm, sb_prev, pr, md.nseg_g, md.Ng, u2_prev, ar.anc_in);
            }
#pragma unroll
            for (int j = 0; j < PG_PPT; ++j)
                if (md.p0 + base_i + slot_of(tid, j) == md.Ng - 1) a[j] = ref_idx;
        }
        PG_STAMP(4);
        // ---- slot-major -> particle-major through LDS, ancestor trace, weight update
        __syncthreads();
#pragma unroll
        for (int j = 0; j < PG_PPT; ++j) {
            sm.u.a[slot_of(tid, j)] = a[j];
            if (base_i + slot_of(tid, j) < N) st_stream(&ar.anc_out[base_i + slot_of(tid, j)], (int32_t)a[j]);
        }
        __syncthreads();
        // log p(y_{t-1} | aux_{t-1}) of the ancestors: this device's row, or the owning peers' (src/PGAS.py:146).  Every slot holds a
        // valid index (also past N), so the four gathers are issued together, unconditionally: a guarded load per particle makes the
        // compiler wait for each in turn (eight dependent round trips instead of one or two).
        int anv[PG_PPT];
        double lav[PG_PPT];
#pragma unroll
        for (int r = 0; r < PG_PPT; ++r) anv[r] = sm.u.a[r * PG_BLK + tid];
        if (pr.world == 1) {   // uniform
            const double* __restrict__ la0 = ar.anc_in.la_s[0];
#pragma unroll
            for (int r = 0; r < PG_PPT; ++r) lav[r] = la0[anv[r]];
        } else {
            const double* rowp[PG_PPT];
            int off[PG_PPT];
#pragma unroll
            for (int r = 0; r < PG_PPT; ++r) {
                const int ra = anv[r] / pr.Nl;
                off[r] = anv[r] - ra * pr.Nl;
                rowp[r] = ar.anc_in.la_s[ra];
            }
#pragma unroll
            for (int r = 0; r < PG_PPT; ++r) lav[r] = rowp[r][off[r]];
        }
#pragma unroll
        for (int r = 0; r < PG_PPT; ++r) {
            const int64_t i = base_i + r * PG_BLK + tid;
            lwp[r] = i < N ? lnv[r] - lav[r] : 0.0;
        }
        PG_STAMP(5);
        if (ar.logw_out != nullptr) {
#pragma unroll
            for (int r = 0; r < PG_PPT; ++r) {
                const int64_t i = base_i + r * PG_BLK + tid;
                if (i < N) ar.logw_out[i] = lwp[r];
            }
        }
        __syncthreads();  // staging area -> ScanSmem reuse
    }
    if (ar.mode & PG_RS_SCAN) {
        double lw[2][PG_PPT];
        {
            // the hand-off rows are padded to whole segments: eight unconditional loads in flight together
            double lat[PG_PPT], ht[PG_PPT];
#pragma unroll
            for (int r = 0; r < PG_PPT; ++r) {
                const size_t pi = (size_t)base_i + r * PG_BLK + tid;
                lat[r] = ld_stream(&ar.la_t[pi]);
                ht[r] = ld_stream(&ar.h_t[pi]);
            }
#pragma unroll
            for (int r = 0; r < PG_PPT; ++r) {
                const size_t pi = (size_t)base_i + r * PG_BLK + tid;
                const bool valid_p = pi < (size_t)N;
                const double l1 = lat[r] + lwp[r];
                lw[0][r] = valid_p ? l1 : -__builtin_inf();
                lw[1][r] = valid_p ? l1 + ht[r] : -__builtin_inf();
            }
        }
        PG_STAMP(6);
        segment_scan<2, false, TAIL>(sm.u.scan, lw, seg, sb_next.nsegp, sb_next.c1, sb_next.c2, sb_next.segk_w, sb_next.segs_w);
        PG_STAMP(8);
        if constexpr (TAIL) {
            if (tid < 64) {   // wave 0: lane 0 stored the partials above
                const int g = seg / PG_GRP;
                const int nb = md.nseg - g * PG_GRP < PG_GRP ? md.nseg - g * PG_GRP : PG_GRP;
                unsigned arrived = 0;
                if (tid == 0) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the write-through stores have left before the counter moves
                    arrived = __hip_atomic_fetch_add(&sb_next.grp_cnt[g], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                arrived = __builtin_amdgcn_readfirstlane(arrived);
                if (arrived == (unsigned)(nb - 1)) {   // this workgroup completed group g: scan it for both CDFs
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    group_records_wave(sb_next, md.nseg, 0, g, true);
                    group_records_wave(sb_next, md.nseg, 1, g, true);
                    if (tid == 0) __hip_atomic_store(&sb_next.grp_cnt[g], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the launch after next
                }
            }
        }
    }
    PG_STAMP(7);
}

// reconstruct_trajectory (src/Filtering.py:40-55) from caller-owned traces and a given final index
__global__ void k_backtrace_idx(int N, int T, int nx, const double* __restrict__ x_trace, const int32_t* __restrict__ anc_trace,
                                int64_t idx, double* __restrict__ traj) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    int64_t b = idx;
    for (int i = T - 1; i >= 0; --i) {
        for (int k = 0; k < nx; ++k) traj[(size_t)i * nx + k] = x_trace[((size_t)i * N + b) * nx + k];
        if (i > 0) b = anc_trace[(size_t)(i - 1) * N + b];
    }
}

// Test hook: the shared arithmetic primitives of include/pgas_detmath.h / pgas_canon.h evaluated on the device, element by element
//   which = 0: exp(x)   1: log(x)   2: sin(pi x), cos(pi x)   3: Philox4x32-10 (x = 6 words per element: counter, key -> 4 words)
//           4: pgas_seg_ref(x)   5: pgas_seg_arg(x, y)   6: pgas_lvl_scale(x, y)   7: standard-normal pair of a Philox block (as 3)
__global__ void k_detmath(int which, const double* __restrict__ x, const double* __restrict__ y, const uint32_t* __restrict__ w,
                          int64_t n, double* __restrict__ o0, double* __restrict__ o1, uint32_t* __restrict__ ow) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    switch (which) {
    case 0: o0[i] = pgas_exp(x[i]); break;
    case 1: o0[i] = pgas_log(x[i]); break;
    case 2: pgas_sincospi(x[i], &o0[i], &o1[i]); break;
    case 3: {
        const pgas_u32x4 r = pgas_philox4x32_10(w[6 * i], w[6 * i + 1], w[6 * i + 2], w[6 * i + 3], w[6 * i + 4], w[6 * i + 5]);
        for (int k = 0; k < 4; ++k) ow[4 * i + k] = r.v[k];
        break;
    }
    case 4: o0[i] = pgas_seg_ref(x[i]); break;
    case 5: o0[i] = pgas_seg_arg(x[i], y[i]); break;
    case 6: o0[i] = pgas_lvl_scale(x[i], y[i]); break;
    case 7: {
        const pgas_u32x4 r = pgas_philox4x32_10(w[6 * i], w[6 * i + 1], w[6 * i + 2], w[6 * i + 3], w[6 * i + 4], w[6 * i + 5]);
        pgas_normal_pair(r, &o0[i], &o1[i]);
        break;
    }
    default: break;
    }
}

// ------------------------------------------------------------------------------------------
// k_sweep_small: the WHOLE sweep (src/PGAS.py:176-228) of a context with at most one segment of particles (N <= 1024: the
// reference's own operating point, N = 200 in src/Toy_Example.py:137 / src/EMPS.py:245) in ONE launch of ONE workgroup.
// The multi-kernel path needs ~3 dependent launches per time step whatever N is (15 us per step at N = 200: launch latency);
// here a step is a handful of workgroup barriers: states and log-weights stay in registers, the two fixed-point CDFs, the
// log-likelihoods of the auxiliary states and the exchange arrays live in LDS, only the traces go to memory.
//
// Same arithmetic as the general path, bit for bit.  With a single segment the hierarchical CDF of DESIGN.md 4.4 collapses:
// group and top references equal the segment's (all scales are exactly 1, all prefixes exactly 0), so
//   num_k = c_k 2^-51,   S = s 2^-51   (c_k the integer cumsum, s its total)
// and the systematic-resampling search, the ancestor draw and the final index are counts #{k : num_k < tau} against that.
// ------------------------------------------------------------------------------------------
// Workgroup barrier that orders LDS only: __syncthreads() also waits for the wave's outstanding global stores (vmcnt(0)), which in the
// single-workgroup sweeps are the trace rows on their way to HBM -- a microsecond per step that nothing in the workgroup waits for.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// The propagation noise of a whole small sweep (src/PGAS.py:72-75), one Philox block + Box-Muller pair per (t, particle), written
// by a grid-wide launch BEFORE the single-workgroup sweep: generated by the lane that owns the particle, in sequence with everything
// else, it is two thirds of the step's latency chain at one wave per SIMD (6 700 of 10 200 cycles, measured); generated ahead by
// noise waves inside the workgroup, the waves got in each other's way (14.5 ms per sweep).  T x N x 16 bytes: 6.4 MB at N = 200.
__global__ __launch_bounds__(256) void k_small_noise(const SweepParams* __restrict__ swp, int64_t p0, int N, int T, double* __restrict__ znoise) {
    const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (q >= (int64_t)N * (T - 1)) return;
    const int t = 1 + (int)(q / N), i = (int)(q % N);
    double z0, z1;
    pgas_normal_pair(pgas_rng_block(ld_const(&swp->seed), PGAS_STREAM_PROP, 0u, (uint32_t)t, (uint64_t)(p0 + i)), &z0, &z1);
    znoise[((size_t)t * N + i) * 2] = z0;
    znoise[((size_t)t * N + i) * 2 + 1] = z1;
}

// one particle through one time step given its noise: transition mean (WIDE: rows of the contraction unrolled), new state, the three
// log-densities -- propagate_group's arithmetic (src/PGAS.py:45-77,90-100,109-116,130-145)
template <int NX, int D, int JIN, int J0T>
__device__ __forceinline__ void small_particle_step(const DevModel& md, const TransParams& tp, const double* __restrict__ G, const double* __restrict__ ut,
                                                    const double* __restrict__ yt, const double (&rf)[NX], bool conditioned, const double (&xin)[1][NX],
                                                    const double (&z)[2], double (&xt)[NX], double& la, double& h, double& ln) {
    double aux[1][NX];
    eval_mean<NX, D, JIN, 1, J0T, true>(md, G, ut, xin, aux);
    la = loglik<NX>(md, yt, aux[0]);
    double quad = 0.0;
#pragma unroll
    for (int k = 0; k < NX; ++k) {
        double w = 0.0;
#pragma unroll
        for (int l = 0; l <= k; ++l) w = PGAS_FMA(tp.LSinv[k * NX + l], rf[l] - aux[0][l], w);
        quad = PGAS_FMA(w, w, quad);
    }
    h = PGAS_FMA(-0.5, quad, tp.cS);
#pragma unroll
    for (int k = 0; k < NX; ++k) {
        double v = aux[0][k];
#pragma unroll
        for (int l = 0; l <= k; ++l) v = PGAS_FMA(tp.LS[k * NX + l], z[l], v);
        xt[k] = conditioned ? rf[k] : v;
    }
    ln = loglik<NX>(md, yt, xt);
}

struct SmallSmem {
    uint64_t q[2][PGAS_SEG];     // numerators in particle order (transposition for the prefix sums)
    double num[2][PGAS_SEG];     // CDF numerators of the resampling / ancestor weights, +inf past N
    double la[PGAS_SEG];         // log p(y_t | aux_t) by particle
    double red[2][PG_BLK / 64];
    uint64_t wtot[2][PG_BLK / 64];
    int cnt[PG_BLK / 64];
};

// fixed-point CDFs of NW weight vectors of the one segment into sm.num[w] (thread -> particles r * 256 + tid, r < NR); returns S per vector.
// NR = particle rows in use (1, 2 or 4: N <= 256, 512, 1024): rows beyond it hold no particle and cost nothing.
template <int NW, int NR>
__device__ __forceinline__ void small_scan(SmallSmem& sm, const double (&lw)[NW][NR], int n, double (&S)[NW]) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        double m = -__builtin_inf();
#pragma unroll
        for (int r = 0; r < NR; ++r) m = __builtin_fmax(m, lw[w][r]);
        m = wave_max(m);
        if (lane == 0) sm.red[w][wave] = m;
    }
    lds_barrier();
    uint64_t qkeep[NW] = {};
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        double m = sm.red[w][0];
#pragma unroll
        for (int v = 1; v < PG_BLK / 64; ++v) m = __builtin_fmax(m, sm.red[w][v]);
        const double kref = pgas_seg_ref(m);
        double arg[NR];
        uint64_t qv[NR];
#pragma unroll
        for (int r = 0; r < NR; ++r) arg[r] = pgas_seg_arg(lw[w][r], kref);
        dev_exp_q51_n<NR>(arg, qv);
        if constexpr (NR == 1) qkeep[w] = qv[0];   // one particle per thread: thread order IS particle order, nothing to transpose
        else {
#pragma unroll
            for (int r = 0; r < NR; ++r) sm.q[w][r * PG_BLK + tid] = qv[r];
        }
    }
    if constexpr (NR > 1) lds_barrier();
    uint64_t loc[NW][NR], incl[NW];
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        uint64_t run = 0;
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            run += NR == 1 ? qkeep[w] : sm.q[w][NR * tid + j];
            loc[w][j] = run;
        }
        incl[w] = wave_incl_scan_u64(run);
        if (lane == 63) sm.wtot[w][wave] = incl[w];
    }
    lds_barrier();
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        uint64_t off = 0, tot = 0;
#pragma unroll
        for (int v = 0; v < PG_BLK / 64; ++v) {
            const uint64_t t = sm.wtot[w][v];
            if (v < wave) off += t;
            tot += t;
        }
        const uint64_t base = off + incl[w] - loc[w][NR - 1];
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            const int k = NR * tid + j;
            sm.num[w][k] = k < n ? pgas_u64_to_double(base + loc[w][j]) * PGAS_FIX_INV : __builtin_inf();
        }
        S[w] = pgas_u64_to_double(tot) * PGAS_FIX_INV;
    }
    lds_barrier();
}

// #{k < span : num[k] < tau} over the +inf padded segment (branch-free lower bound; span = NR * 256, a power of two)
template <int SPAN>
__device__ __forceinline__ int small_lower_bound(const double* __restrict__ num, double tau) {
    int p = 0;
#pragma unroll
    for (int step = SPAN / 2; step >= 1; step >>= 1)
        if (num[p + step - 1] < tau) p += step;
    return (num[p] < tau) ? p + 1 : p;   // the element the halving never looks at: index SPAN - 1
}

// the same count for ONE threshold by the whole workgroup (ancestor of the conditioned particle, final index)
template <int NR>
__device__ __forceinline__ int small_count(SmallSmem& sm, const double* __restrict__ num, double tau) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int k = 0;
#pragma unroll
    for (int j = 0; j < NR; ++j) k += (num[j * PG_BLK + tid] < tau) ? 1 : 0;
    k = wave_sum_i(k);
    if (lane == 0) sm.cnt[wave] = k;
    lds_barrier();
    int tot = 0;
#pragma unroll
    for (int v = 0; v < PG_BLK / 64; ++v) tot += sm.cnt[v];
    lds_barrier();
    return tot;
}

template <int NX, int D, int JIN, int J0T, int NR>
__global__ __launch_bounds__(PG_BLK) void k_sweep_small(DevModel md, const TransParams* __restrict__ tpp, const double* __restrict__ G_arg,
                                                        const SweepParams* __restrict__ swp, const double* __restrict__ u_res,
                                                        const double* __restrict__ u_anc, const double* __restrict__ m0L0,
                                                        const double* __restrict__ ref, double* __restrict__ x_trace,
                                                        int32_t* __restrict__ anc_trace, double* __restrict__ logw_last,
                                                        double* __restrict__ logw_trace /* (T, N) or NULL */, UpperHdr* __restrict__ hdr,
                                                        double* __restrict__ traj, const double* __restrict__ znoise /* (T, N, 2): k_small_noise */) {
    __shared__ SmallSmem sm;
    extern __shared__ __attribute__((aligned(16))) double pg_g_lds_small[];   // the coefficient tensor (every basis shape: one wave per SIMD
                                                                               // has nothing to hide a scalar load per grid row behind)
    const int tid = threadIdx.x;
    const int N = md.N, T = md.T;   // particle i = r * 256 + tid, r < NR
    const size_t row = (size_t)N * NX;
    TransParams tp;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        tp.LS[q] = ld_const(&tpp->LS[q]);
        tp.LSinv[q] = ld_const(&tpp->LSinv[q]);
    }
    tp.cS = ld_const(&tpp->cS);
    tp.G = G_arg;
    const uint64_t seed = ld_const(&swp->seed);
    {
        int gtot = NX;
#pragma unroll
        for (int d = 0; d < D; ++d) gtot *= (d == D - 1 && D > 1) ? JIN : md.J[d];
        for (int i = tid; i < gtot; i += PG_BLK) pg_g_lds_small[i] = G_arg[i];
        lds_barrier();
    }
    const double* Guse = pg_g_lds_small;
    const bool pow2 = (N & (N - 1)) == 0;
    const double invN = 1.0 / (double)N;

    // ---- x_0 ~ N(m0, P0), conditioned particle = ref_0 (src/PGAS.py:155-174,194)
    double x[NR][NX], logw[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int i = r * PG_BLK + tid;
        logw[r] = 0.0;
        double z[2];
        pgas_rng_normals(seed, PGAS_STREAM_INIT, 0u, (uint64_t)(i < N ? i : N - 1), NX, z);
#pragma unroll
        for (int k = 0; k < NX; ++k) {
            double v = m0L0[k];
#pragma unroll
            for (int l = 0; l <= k; ++l) v = PGAS_FMA(m0L0[NX + k * NX + l], z[l], v);
            x[r][k] = (i == N - 1) ? ref[k] : v;
        }
        if (i < N) {
#pragma unroll
            for (int k = 0; k < NX; ++k) x_trace[(size_t)i * NX + k] = x[r][k];
            if (logw_trace != nullptr) logw_trace[i] = 0.0;
        }
    }

    // ---- the time loop (src/PGAS.py:199-221).  y_t, ref_t, the uniforms and the particles' noise are fetched one step ahead.
    double yn[PGAS_MAX_NY], rn[NX], u1n = 0.0, u2n = 0.0;
    double2 zn[NR];
    auto fetch = [&](int t) {
#pragma unroll
        for (int k = 0; k < PGAS_MAX_NY; ++k) yn[k] = k < md.ny ? md.y[(size_t)t * md.ny + k] : 0.0;
#pragma unroll
        for (int k = 0; k < NX; ++k) rn[k] = ref[(size_t)t * NX + k];
        u1n = u_res[t];
        u2n = u_anc[t];
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int i = r * PG_BLK + tid;
            zn[r] = reinterpret_cast<const double2*>(znoise)[(size_t)t * N + (i < N ? i : N - 1)];
        }
    };
    if (T > 1) fetch(1);
    for (int t = 1; t < T; ++t) {
        const double* __restrict__ ut = md.u + (size_t)t * md.nu;
        double yt[PGAS_MAX_NY], rf[NX];
#pragma unroll
        for (int k = 0; k < PGAS_MAX_NY; ++k) yt[k] = yn[k];
#pragma unroll
        for (int k = 0; k < NX; ++k) rf[k] = rn[k];
        const double u1 = u1n, u2 = u2n;
        double2 zc[NR];
#pragma unroll
        for (int r = 0; r < NR; ++r) zc[r] = zn[r];
        fetch(t + 1 < T ? t + 1 : t);
        double lw[2][NR], ln[NR];
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int i = r * PG_BLK + tid;
            double xin[1][NX], xt[NX], la1, h1, ln1;
#pragma unroll
            for (int k = 0; k < NX; ++k) xin[0][k] = x[r][k];
            const double z[2] = {zc[r].x, zc[r].y};
            small_particle_step<NX, D, JIN, J0T>(md, tp, Guse, ut, yt, rf, md.p0 + i == md.Ng - 1, xin, z, xt, la1, h1, ln1);
            lw[0][r] = -__builtin_inf();
            lw[1][r] = -__builtin_inf();
            ln[r] = ln1;
            if (i < N) {
#pragma unroll
                for (int k = 0; k < NX; ++k) {
                    x[r][k] = xt[k];
                    st_stream(&x_trace[(size_t)t * row + (size_t)i * NX + k], xt[k]);
                }
                const double l1 = la1 + logw[r];   // src/PGAS.py:101-102
                lw[0][r] = l1;
                lw[1][r] = l1 + h1;                // :117-118
            }
            sm.la[i] = la1;
        }
#ifdef PG_STAMPS
#define PG_SSTAMP(k) do { if (tid == 0 && (t == 100 || t == 101)) g_stamps[(t - 100) * 8 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
        PG_SSTAMP(0);
        lds_barrier();   // diagnostic build only: a phase boundary after the propagation
        PG_SSTAMP(1);
#else
#define PG_SSTAMP(k) do { } while (0)
#endif
        double S[2];
        small_scan<2, NR>(sm, lw, N, S);   // ends with a barrier: sm.num and sm.la are visible
        PG_SSTAMP(2);
        // ---- systematic resampling (src/Filtering.py:28-35) and the ancestor of the conditioned particle (src/PGAS.py:121-127)
        const bool valid1 = (S[0] > 0.0) && (S[0] < __builtin_inf()), valid2 = (S[1] > 0.0) && (S[1] < __builtin_inf());
        const int cnt2 = small_count<NR>(sm, sm.num[1], u2 * S[1]);
        const int ref_idx = valid2 ? (cnt2 > N - 1 ? N - 1 : cnt2) : N - 1;
        PG_SSTAMP(3);
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int i = r * PG_BLK + tid;
            if (i < N) {
                int a = i;   // no positive weight: identity (src/Filtering.py:25)
                if (valid1) {
                    const int p = small_lower_bound<NR * PG_BLK>(sm.num[0], slot_U(u1, i, N, invN, pow2) * S[0]);
                    a = p > N - 1 ? N - 1 : p;
                }
                if (i == N - 1) a = ref_idx;
                anc_trace[(size_t)(t - 1) * N + i] = (int32_t)a;   // plain store: the back-trace of this very launch reads it (L2)
                logw[r] = ln[r] - sm.la[a];   // src/PGAS.py:137-147
                if (logw_trace != nullptr) logw_trace[(size_t)t * N + i] = logw[r];
            }
        }
        lds_barrier();   // sm.la / sm.num are rewritten by the next step
        PG_SSTAMP(4);
    }

    // ---- final index (src/PGAS.py:224-225) and back-trace (src/Filtering.py:40-55)
    double lwf[1][NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int i = r * PG_BLK + tid;
        lwf[0][r] = i < N ? logw[r] : -__builtin_inf();
        if (i < N) logw_last[i] = logw[r];
    }
    double Sf[1];
    small_scan<1, NR>(sm, lwf, N, Sf);
    const bool validf = (Sf[0] > 0.0) && (Sf[0] < __builtin_inf());
    const int cf = small_count<NR>(sm, sm.num[0], ld_const(&swp->u_final) * Sf[0]);
    const int fidx = validf ? (cf > N - 1 ? N - 1 : cf) : N - 1;
    // the traces were written by every wave of this workgroup: make them visible to the one lane that chases
    __threadfence();
    lds_barrier();
    if (tid == 0) {
        hdr->final_idx = fidx;
        int b = fidx;
#ifdef PG_SMALL_NO_BT
        if (T > 0) return;
#endif
        for (int i = T - 1; i >= 0; --i) {
#pragma unroll
            for (int k = 0; k < NX; ++k) traj[(size_t)i * NX + k] = __hip_atomic_load(&x_trace[(size_t)i * row + (size_t)b * NX + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (i > 0) b = __hip_atomic_load(&anc_trace[(size_t)(i - 1) * N + b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// ------------------------------------------------------------------------------------------
// k_sweep_duo: the small sweep (N <= 1024) on TWO workgroups -- the structure of the large sweep in
// miniature.  Quirk Q1: x_t[i] depends on x_{t-1}[i] only, so workgroup 0 runs the propagation of every step ahead and leaves
// (log p(y_t|aux_t), log N(ref_t; aux_t, S), log p(y_t|x_t)) of every particle in a ring in device memory; workgroup 1 runs the
// weight recursion (both softmax scans, resampling search, ancestor draw, weight update) one ring slot behind, then the final index
// and the back-trace.  In k_sweep_small the two halves add up inside every step (3 100 + 4 200 of 7 300 cycles at N = 200); here they
// overlap.  Inside ONE workgroup they cannot: a workgroup barrier is for all of its waves, and the recursion needs seven per step.
// The two meet through two counters in device memory (agent-scope release / acquire); the propagation never waits for the
// recursion except for ring space, so with both workgroups resident -- a grid of two -- the pipeline cannot deadlock.
// ------------------------------------------------------------------------------------------
#define PG_DUO_RING 16
struct DuoShared {   // device memory, one per context
    int ready;       // propagation has delivered steps <= ready (T when it has also fenced its trace stores)
    int done;        // the recursion has consumed steps <= done
    int pad[14];
    double ring[PG_DUO_RING][3][PGAS_SEG];
};
__device__ __forceinline__ int duo_poll(const int* p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT); }

// Lineage checkpoints of the recursion workgroup (N <= 1024: indices fit 16 bits).  The back-trace is T - 1 dependent loads, 0.35 us
// each out of L2: 0.7 ms of a 5 ms sweep when one lane chases them in sequence.  The recursion therefore tracks, for every current
// particle, the index of its ancestor at the last checkpoint time (one LDS gather per step) and files that map away every
// C = ceil((T-1)/32) steps; after the final index is known, at most 32 LDS look-ups give the lineage's index at every checkpoint and
// the segments in between are chased by 32 lanes in parallel.
#define PG_DUO_CP 32
struct DuoLineage {
    uint16_t root[2][PGAS_SEG];          // step parity: index at the last checkpoint time of the lineage of particle i
    uint16_t cp[PG_DUO_CP][PGAS_SEG];    // cp[k][i]: index at time k C of the lineage of particle i at time (k + 1) C
    int at[PG_DUO_CP + 1];               // the sampled lineage's index at time k C
};

template <int NX, int D, int JIN, int J0T, int NR>
__global__ __launch_bounds__(PG_BLK) void k_sweep_duo(DevModel md, const TransParams* __restrict__ tpp, const double* __restrict__ G_arg,
                                                      const SweepParams* __restrict__ swp, const double* __restrict__ u_res,
                                                      const double* __restrict__ u_anc, const double* __restrict__ m0L0,
                                                      const double* __restrict__ ref, double* __restrict__ x_trace,
                                                      int32_t* __restrict__ anc_trace, double* __restrict__ logw_last,
                                                      double* __restrict__ logw_trace /* (T, N) or NULL */, UpperHdr* __restrict__ hdr,
                                                      double* __restrict__ traj, const double* __restrict__ znoise, DuoShared* __restrict__ duo) {
    __shared__ SmallSmem sm;
    __shared__ DuoLineage lin;
    extern __shared__ __attribute__((aligned(16))) double pg_g_lds_duo[];
    const int tid = threadIdx.x;
    const int N = md.N, T = md.T;
    const size_t row = (size_t)N * NX;
    const int C = T > 1 ? (T - 1 + PG_DUO_CP - 1) / PG_DUO_CP : 1;   // checkpoint interval: at most PG_DUO_CP blocks of steps
    if (blockIdx.x == 0) {
        // ================= workgroup 0: propagation (src/PGAS.py:45-77,130-134), particle i = r * 256 + tid
        TransParams tp;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            tp.LS[q] = ld_const(&tpp->LS[q]);
            tp.LSinv[q] = ld_const(&tpp->LSinv[q]);
        }
        tp.cS = ld_const(&tpp->cS);
        tp.G = G_arg;
        const uint64_t seed = ld_const(&swp->seed);
        {
            int gtot = NX;
#pragma unroll
            for (int d = 0; d < D; ++d) gtot *= (d == D - 1 && D > 1) ? JIN : md.J[d];
            for (int i = tid; i < gtot; i += PG_BLK) pg_g_lds_duo[i] = G_arg[i];
            lds_barrier();
        }
        double x[NR][NX];
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int i = r * PG_BLK + tid;
            double z[2];
            pgas_rng_normals(seed, PGAS_STREAM_INIT, 0u, (uint64_t)(i < N ? i : N - 1), NX, z);
#pragma unroll
            for (int k = 0; k < NX; ++k) {
                double v = m0L0[k];
#pragma unroll
                for (int l = 0; l <= k; ++l) v = PGAS_FMA(m0L0[NX + k * NX + l], z[l], v);
                x[r][k] = (i == N - 1) ? ref[k] : v;
                if (i < N) x_trace[(size_t)i * NX + k] = x[r][k];
            }
        }
        double yn[PGAS_MAX_NY], rn[NX];
        double2 zn[NR];
        auto fetch = [&](int t) {
#pragma unroll
            for (int k = 0; k < PGAS_MAX_NY; ++k) yn[k] = k < md.ny ? md.y[(size_t)t * md.ny + k] : 0.0;
#pragma unroll
            for (int k = 0; k < NX; ++k) rn[k] = ref[(size_t)t * NX + k];
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                const int i = r * PG_BLK + tid;
                zn[r] = reinterpret_cast<const double2*>(znoise)[(size_t)t * N + (i < N ? i : N - 1)];
            }
        };
        if (T > 1) fetch(1);
        // A step is PUBLISHED one iteration late: by then its ring stores (and the trace stores in front of them in the in-order
        // vmcnt queue) have had a whole step to complete, and the wait for them costs next to nothing.
        auto publish = [&](int t) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every wave's slot stores have left
            lds_barrier();
            if (tid == 0) __hip_atomic_store(&duo->ready, t, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        };
        for (int t = 1; t < T; ++t) {
            const double* __restrict__ ut = md.u + (size_t)t * md.nu;
            double yt[PGAS_MAX_NY], rf[NX];
#pragma unroll
            for (int k = 0; k < PGAS_MAX_NY; ++k) yt[k] = yn[k];
#pragma unroll
            for (int k = 0; k < NX; ++k) rf[k] = rn[k];
            double2 zc[NR];
#pragma unroll
            for (int r = 0; r < NR; ++r) zc[r] = zn[r];
            if (t > 1) publish(t - 1);
            fetch(t + 1 < T ? t + 1 : t);
            double la[NR], h[NR], ln[NR];
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                const int i = r * PG_BLK + tid;
                double xin[1][NX], xt[NX];
#pragma unroll
                for (int k = 0; k < NX; ++k) xin[0][k] = x[r][k];
                const double z[2] = {zc[r].x, zc[r].y};
                small_particle_step<NX, D, JIN, J0T>(md, tp, pg_g_lds_duo, ut, yt, rf, md.p0 + i == md.Ng - 1, xin, z, xt, la[r], h[r], ln[r]);
                if (i < N) {
#pragma unroll
                    for (int k = 0; k < NX; ++k) {
                        x[r][k] = xt[k];
                        st_stream(&x_trace[(size_t)t * row + (size_t)i * NX + k], xt[k]);
                    }
                }
            }
            // the slot of step t is free once the recursion has consumed step t - RING (one wave polls for all)
            if (t > PG_DUO_RING) {
                if (tid == 0)
                    while (duo_poll(&duo->done) < t - PG_DUO_RING) __builtin_amdgcn_s_sleep(2);
                lds_barrier();
            }
            double (&slot)[3][PGAS_SEG] = duo->ring[t & (PG_DUO_RING - 1)];
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                const int i = r * PG_BLK + tid;
                __hip_atomic_store(&slot[0][i], la[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&slot[1][i], h[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&slot[2][i], ln[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        if (T > 1) publish(T - 1);
        __threadfence();   // the state trace is in memory before the other workgroup chases ancestors through it
        lds_barrier();
        if (tid == 0) __hip_atomic_store(&duo->ready, T, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    // ================= workgroup 1: the weight recursion (src/PGAS.py:90-127,137-147), final index, back-trace
    const bool pow2 = (N & (N - 1)) == 0;
    const double invN = 1.0 / (double)N;
    double logw[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        logw[r] = 0.0;
        if (logw_trace != nullptr && r * PG_BLK + tid < N) logw_trace[r * PG_BLK + tid] = 0.0;
    }
    double u1n = T > 1 ? u_res[1] : 0.0, u2n = T > 1 ? u_anc[1] : 0.0;
    int seen = 0;            // last value of the propagation's counter this workgroup has read: it runs ahead, so most steps need no poll
    bool have_next = false;  // the next step's slot is already in registers (fetched while this step ran)
    double nla[NR], nh[NR], nln[NR];
    auto load_slot = [&](int t) {
        const double (&slot)[3][PGAS_SEG] = duo->ring[t & (PG_DUO_RING - 1)];
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int i = r * PG_BLK + tid;
            nla[r] = __hip_atomic_load(&slot[0][i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            nh[r] = __hip_atomic_load(&slot[1][i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            nln[r] = __hip_atomic_load(&slot[2][i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    };
    for (int t = 1; t < T; ++t) {
        const double u1 = u1n, u2 = u2n;
        u1n = u_res[t + 1 < T ? t + 1 : t];
        u2n = u_anc[t + 1 < T ? t + 1 : t];
        double U[NR];   // the data-independent half of the thresholds (a division when N is no power of two) before the wait
#pragma unroll
        for (int r = 0; r < NR; ++r) U[r] = slot_U(u1, r * PG_BLK + tid, N, invN, pow2);
        if (!have_next) {
            if (seen < t) {   // uniform
                if (tid == 0) {
                    int v;
                    while ((v = duo_poll(&duo->ready)) < t) __builtin_amdgcn_s_sleep(1);
                    sm.cnt[0] = v;
                }
                lds_barrier();
                seen = sm.cnt[0];
                lds_barrier();   // sm.cnt is reused by small_count
            }
            load_slot(t);
        }
        double lw[2][NR], ln[NR];
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int i = r * PG_BLK + tid;
            const double la = nla[r], h = nh[r];
            ln[r] = nln[r];
            const double l1 = la + logw[r];   // src/PGAS.py:101-102
            lw[0][r] = i < N ? l1 : -__builtin_inf();
            lw[1][r] = i < N ? l1 + h : -__builtin_inf();   // :117-118
            sm.la[i] = la;
        }
        // the next step's slot, if the propagation has already delivered it: its latency hides behind this step's scans
        have_next = t + 1 < T && seen >= t + 1;
        if (have_next) load_slot(t + 1);
        double S[2];
        small_scan<2, NR>(sm, lw, N, S);   // ends with a barrier: sm.num and sm.la are visible
        const bool valid1 = (S[0] > 0.0) && (S[0] < __builtin_inf()), valid2 = (S[1] > 0.0) && (S[1] < __builtin_inf());
        const int cnt2 = small_count<NR>(sm, sm.num[1], u2 * S[1]);
        const int ref_idx = valid2 ? (cnt2 > N - 1 ? N - 1 : cnt2) : N - 1;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int i = r * PG_BLK + tid;
            if (i < N) {
                int a = i;   // no positive weight: identity (src/Filtering.py:25)
                if (valid1) {
                    const int p = small_lower_bound<NR * PG_BLK>(sm.num[0], U[r] * S[0]);
                    a = p > N - 1 ? N - 1 : p;
                }
                if (i == N - 1) a = ref_idx;
                anc_trace[(size_t)(t - 1) * N + i] = (int32_t)a;   // plain store: the back-trace of this very launch reads it (L2)
                logw[r] = ln[r] - sm.la[a];   // src/PGAS.py:137-147
                if (logw_trace != nullptr) logw_trace[(size_t)t * N + i] = logw[r];
                // lineage: index at the last checkpoint time (k C, k = (t-1) / C) of particle i's ancestor
                const uint16_t rt = ((t - 1) % C == 0) ? (uint16_t)a : lin.root[(t - 1) & 1][a];
                lin.root[t & 1][i] = rt;
                if (t % C == 0) lin.cp[t / C - 1][i] = rt;
            }
        }
        lds_barrier();   // sm.la / sm.num are rewritten by the next step; every wave has read its ring slot
        if (tid == 0) __hip_atomic_store(&duo->done, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // ---- final index (src/PGAS.py:224-225) and back-trace (src/Filtering.py:40-55)
    double lwf[1][NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int i = r * PG_BLK + tid;
        lwf[0][r] = i < N ? logw[r] : -__builtin_inf();
        if (i < N) logw_last[i] = logw[r];
    }
    double Sf[1];
    small_scan<1, NR>(sm, lwf, N, Sf);
    const bool validf = (Sf[0] > 0.0) && (Sf[0] < __builtin_inf());
    const int cf = small_count<NR>(sm, sm.num[0], ld_const(&swp->u_final) * Sf[0]);
    const int fidx = validf ? (cf > N - 1 ? N - 1 : cf) : N - 1;
    __threadfence();   // this workgroup's ancestor stores
    if (tid == 0) {
        while (duo_poll(&duo->ready) < T) __builtin_amdgcn_s_sleep(1);   // the propagation has written and fenced the state trace
        hdr->final_idx = fidx;
        // the lineage's index at every checkpoint time, newest first: the running map covers the last (partial) block
        const int kl = T > 1 ? (T - 2) / C : 0;   // block of the last step T - 1
        lin.at[kl + 1] = fidx;
        int b = T > 1 ? (int)lin.root[(T - 1) & 1][fidx] : fidx;
        lin.at[kl] = b;
        for (int k = kl - 1; k >= 0; --k) {
            b = lin.cp[k][b];
            lin.at[k] = b;
        }
    }
    lds_barrier();
    {
        // segment k = times (k C, top], top = min((k + 1) C, T - 1), chased by lane k from the lineage's index at `top`
        const int kl = T > 1 ? (T - 2) / C : 0;
        if (tid <= kl && T > 1) {
            const int k = tid, top = k == kl ? T - 1 : (k + 1) * C;
            int b = lin.at[k + 1];
            for (int t = top; t > k * C; --t) {
#pragma unroll
                for (int q = 0; q < NX; ++q) traj[(size_t)t * NX + q] = __hip_atomic_load(&x_trace[(size_t)t * row + (size_t)b * NX + q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                b = __hip_atomic_load(&anc_trace[(size_t)(t - 1) * N + b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (k == 0) {
#pragma unroll
                for (int q = 0; q < NX; ++q) traj[q] = __hip_atomic_load(&x_trace[(size_t)b * NX + q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        if (T == 1 && tid == 0) {
#pragma unroll
            for (int q = 0; q < NX; ++q) traj[q] = __hip_atomic_load(&x_trace[(size_t)fidx * NX + q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}
