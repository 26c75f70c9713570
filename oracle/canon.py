"""ctypes front end of the canonical C oracle (oracle/pgas_canon.c).

TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py cpu_baseline).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libpgas_oracle.so")

STREAM_INIT, STREAM_PROP, STREAM_RESAMPLE, STREAM_ANCESTOR, STREAM_FINAL = 1, 2, 3, 4, 5
STREAM_M_INIT_STATE, STREAM_M_STATE, STREAM_M_RESAMPLE, STREAM_M_ANCESTOR, STREAM_M_FINAL, STREAM_M_INIT_INTVAR, STREAM_M_INTVAR = 16, 17, 18, 19, 20, 24, 32


def build(force=False):
    src = os.path.join(_HERE, "pgas_canon.c")
    deps = [src, os.path.join(_HERE, "..", "include", "pgas_canon.h"), os.path.join(_HERE, "..", "include", "pgas_detmath.h")]
    if force or not os.path.exists(_SO) or (
        all(os.path.exists(d) for d in deps) and os.path.getmtime(_SO) < max(os.path.getmtime(d) for d in deps)
    ):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libpgas_oracle.so"], stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        dp, ip, u64p = C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_uint64)
        L.oc_model_create.restype = C.c_void_p
        L.oc_model_create.argtypes = [C.c_int32] * 7 + [ip, ip, dp, dp, C.c_double, dp, dp, C.c_double, dp, dp]
        L.oc_model_destroy.argtypes = [C.c_void_p]
        L.oc_model_grid.argtypes = [C.c_void_p, ip, ip, ip]
        L.oc_grid_size.restype = C.c_int64
        L.oc_grid_size.argtypes = [C.c_void_p]
        L.oc_pack_coeff.argtypes = [C.c_void_p, dp, dp]
        L.oc_basis_eval.argtypes = [C.c_void_p, dp, C.c_int32, C.c_int64, dp]
        L.oc_step.restype = C.c_int
        L.oc_step.argtypes = [C.c_void_p, C.c_int32, C.c_uint64, dp, dp, dp, dp, dp, C.c_double, dp, dp, dp, ip, dp, dp, dp, dp, dp]
        L.oc_init_state.argtypes = [C.c_void_p, C.c_uint64, dp, dp, dp, dp]
        L.oc_final_index.restype = C.c_int64
        L.oc_final_index.argtypes = [C.c_void_p, C.c_uint64, dp]
        L.oc_set_corrected.restype = None
        L.oc_set_corrected.argtypes = [C.c_void_p, C.c_int32]
        L.oc_sweep.restype = C.c_int
        L.oc_sweep.argtypes = [C.c_void_p, C.c_uint64, dp, dp, dp, dp, C.c_double, dp, dp, dp, dp, ip, dp, C.c_int32]
        L.oc_exp_v.argtypes = [dp, dp, C.c_int64]
        L.oc_log_v.argtypes = [dp, dp, C.c_int64]
        L.oc_sincospi_v.argtypes = [dp, dp, dp, C.c_int64]
        L.oc_philox.argtypes = [C.POINTER(C.c_uint32)] * 3
        L.oc_uniform.restype = C.c_double
        L.oc_uniform.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32]
        L.oc_normals.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_int64, C.c_int64, C.c_int, dp]
        L.oc_chi2.restype = None
        L.oc_chi2.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_int64, C.c_int64, dp, dp]
        L.oc_student_t.restype = None
        L.oc_student_t.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_int64, C.c_int64, dp, dp]
        L.oc_gamma.restype = None
        L.oc_gamma.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_int64, C.c_int64, dp, dp]
        L.oc_u64_to_double.restype = C.c_double
        L.oc_u64_to_double.argtypes = [C.c_uint64]
        L.oc_segment_partials.argtypes = [dp, C.c_int64, dp, u64p, u64p]
        L.oc_resample_range.argtypes = [C.c_int32, dp, u64p, u64p, C.c_int64, C.c_double, C.c_int64, C.c_int64, ip]
        L.oc_seg.restype = C.c_int32
        _lib = L
    return _lib


def _dp(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_int32))


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


# ---------------------------------------------------------------- primitive hooks
def det_exp(x):
    x = _f64(x)
    out = np.empty_like(x)
    lib().oc_exp_v(_dp(x), _dp(out), x.size)
    return out


def det_log(x):
    x = _f64(x)
    out = np.empty_like(x)
    lib().oc_log_v(_dp(x), _dp(out), x.size)
    return out


def det_sincospi(x):
    x = _f64(x)
    s, c = np.empty_like(x), np.empty_like(x)
    lib().oc_sincospi_v(_dp(x), _dp(s), _dp(c), x.size)
    return s, c


def philox(ctr, key):
    c = (C.c_uint32 * 4)(*ctr)
    k = (C.c_uint32 * 2)(*key)
    o = (C.c_uint32 * 4)()
    lib().oc_philox(c, k, o)
    return list(o)


def uniform(seed, stream, t):
    return lib().oc_uniform(seed, stream, t)


def normals(seed, stream, t, p0, n_particles, n):
    z = np.empty((n_particles, n))
    lib().oc_normals(seed, stream, t, p0, n_particles, n, _dp(z))
    return z


def chi2(seed, stream, t, p0, nu):
    """chi^2(nu_p) on the counters of particle p0 + p (2 x pgas_rng_gamma(nu_p / 2))."""
    nu = np.ascontiguousarray(nu, dtype=np.float64)
    out = np.empty_like(nu)
    lib().oc_chi2(seed, stream, t, p0, nu.size, _dp(nu), _dp(out))
    return out


def student_t(seed, stream, t, p0, nu):
    """Student-t(nu[p]) variates of particles p0.. (canonical Marsaglia-Tsang gamma sampler, include/pgas_canon.h)."""
    nu = _f64(nu)
    out = np.empty(nu.size)
    lib().oc_student_t(seed, stream, t, p0, nu.size, _dp(nu), _dp(out))
    return out


def gamma(seed, stream, t, p0, a):
    a = _f64(a)
    out = np.empty(a.size)
    lib().oc_gamma(seed, stream, t, p0, a.size, _dp(a), _dp(out))
    return out


def segment_partials(lw):
    """Per-segment softmax partials of a log-weight vector: (segm (nseg,), segs (nseg,) uint64, c (n,) uint64)."""
    lw = _f64(lw)
    n = lw.size
    nseg = (n + 1023) // 1024
    segm, segs, c = np.empty(nseg), np.empty(nseg, np.uint64), np.empty(n, np.uint64)
    u64p = C.POINTER(C.c_uint64)
    lib().oc_segment_partials(_dp(lw), n, _dp(segm), segs.ctypes.data_as(u64p), c.ctypes.data_as(u64p))
    return segm, segs, c


def resample_range(segm, segs, c, N, u, i0, i1):
    """Ancestors of slots [i0, i1) given ALL segments' partials and cumsums (what a rank holds after the all-gather)."""
    segm, segs, c = _f64(segm), np.ascontiguousarray(segs, np.uint64), np.ascontiguousarray(c, np.uint64)
    anc = np.empty(i1 - i0, np.int32)
    u64p = C.POINTER(C.c_uint64)
    lib().oc_resample_range(len(segm), _dp(segm), segs.ctypes.data_as(u64p), c.ctypes.data_as(u64p), N, u, i0, i1, _ip(anc))
    return anc


# ---------------------------------------------------------------- model + algorithm
class CanonModel:
    """Holds the same declarative model description the HIP engine takes (see include/pgas_hip.h)."""

    def __init__(self, N, T, nx, ny, nu, idx, sel, alpha, beta, nrm, H, LRinv, cR, y, u):
        idx = np.ascontiguousarray(np.atleast_2d(idx), dtype=np.int32)
        self.N, self.T, self.nx, self.ny, self.nu = N, T, nx, ny, nu
        self.M, self.D = idx.shape
        self.idx = idx
        self.sel = np.ascontiguousarray(sel, dtype=np.int32)
        self.alpha, self.beta, self.nrm = _f64(alpha), _f64(beta), float(nrm)
        self.H, self.LRinv, self.cR = _f64(H).reshape(ny, nx), _f64(LRinv).reshape(ny, ny), float(cR)
        self.y = _f64(y).reshape(T, ny)
        self.u = _f64(u).reshape(T, nu)
        ubuf = self.u if nu else np.zeros(1)
        self._h = lib().oc_model_create(
            N, T, nx, ny, nu, self.D, self.M, _ip(self.idx), _ip(self.sel), _dp(self.alpha), _dp(self.beta),
            self.nrm, _dp(self.H), _dp(self.LRinv), self.cR, _dp(self.y), _dp(ubuf),
        )
        if not self._h:
            raise ValueError("oc_model_create rejected the model description")

    def set_corrected(self, on):
        """CORRECTED mode (resample before propagate, quirk Q1 removed); default off = the reference's behaviour."""
        lib().oc_set_corrected(self._h, 1 if on else 0)

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                lib().oc_model_destroy(self._h)
                self._h = None
        except Exception:  # interpreter shutdown
            pass

    def grid(self):
        J, j0, js = (np.zeros(self.D, np.int32) for _ in range(3))
        lib().oc_model_grid(self._h, _ip(J), _ip(j0), _ip(js))
        return J, j0, js

    def pack_coeff(self, A):
        A = _f64(A).reshape(self.nx, self.M)
        G = np.empty(self.nx * lib().oc_grid_size(self._h))
        lib().oc_pack_coeff(self._h, _dp(A), _dp(G))
        return G

    def basis_eval(self, x, t):
        x = _f64(x).reshape(-1, self.nx)
        phi = np.empty((x.shape[0], self.M))
        lib().oc_basis_eval(self._h, _dp(x), t, x.shape[0], _dp(phi))
        return phi

    @staticmethod
    def chol_parts(S):
        S = np.atleast_2d(np.asarray(S, dtype=np.float64))
        LS = np.linalg.cholesky(S)
        LSinv = np.linalg.inv(LS)
        cS = -0.5 * S.shape[0] * np.log(2 * np.pi) - np.sum(np.log(np.diag(LS)))
        return _f64(LS), _f64(LSinv), float(cS)

    @staticmethod
    def chol_parts_dev(S):
        """The factorisation pgas_set_params_dev performs on the device (csrc/pgas_kernels.hip.h, k_pack), operation for operation in
        IEEE double arithmetic: what a chain that keeps error_cov on the device runs its sweeps with (nx <= 2)."""
        S = np.atleast_2d(np.asarray(S, dtype=np.float64))
        nx = S.shape[0]
        LS, LSinv = np.zeros((nx, nx)), np.zeros((nx, nx))
        log2pi = np.float64(float.fromhex("0x1.d67f1c864beb4p+0"))
        if nx == 1:
            l = np.sqrt(S[0, 0])
            LS[0, 0], LSinv[0, 0] = l, np.float64(1.0) / l
            cS = np.float64(-0.5) * log2pi - det_log(np.array([l]))[0]
        elif nx == 2:
            l00 = np.sqrt(S[0, 0])
            l10 = S[1, 0] / l00
            l11 = np.sqrt(S[1, 1] - l10 * l10)
            i00, i11 = np.float64(1.0) / l00, np.float64(1.0) / l11
            LS[0, 0], LS[1, 0], LS[1, 1] = l00, l10, l11
            LSinv[0, 0], LSinv[1, 0], LSinv[1, 1] = i00, -(l10 * i00) * i11, i11
            cS = -log2pi - (det_log(np.array([l00]))[0] + det_log(np.array([l11]))[0])
        else:
            raise ValueError("chol_parts_dev: nx <= 2")
        return _f64(LS), _f64(LSinv), float(cS)

    def step(self, t, seed, x_prev, logw_prev, A, LS, LSinv, cS, ref_t, debug=False):
        N, nx = self.N, self.nx
        x_prev = _f64(x_prev).reshape(N, nx)
        lwp = None if logw_prev is None else _f64(logw_prev)
        A = _f64(A).reshape(nx, self.M)
        LS, LSinv, ref_t = _f64(LS), _f64(LSinv), _f64(ref_t).reshape(nx)
        lw, xn, anc = np.empty(N), np.empty((N, nx)), np.empty(N, np.int32)
        dbg = {}
        if debug:
            dbg = dict(laux=np.empty(N), lw1=np.empty(N), lw2=np.empty(N), aux=np.empty((N, nx)), u=np.empty(4))
        rc = lib().oc_step(
            self._h, t, seed, _dp(x_prev), _dp(lwp), _dp(A), _dp(LS), _dp(LSinv), cS, _dp(ref_t), _dp(lw), _dp(xn),
            _ip(anc), _dp(dbg.get("laux")), _dp(dbg.get("lw1")), _dp(dbg.get("lw2")), _dp(dbg.get("aux")), _dp(dbg.get("u")),
        )
        assert rc == 0
        return (lw, xn, anc, dbg) if debug else (lw, xn, anc)

    def init_state(self, seed, m0, L0, ref0):
        x0 = np.empty((self.N, self.nx))
        lib().oc_init_state(self._h, seed, _dp(_f64(m0)), _dp(_f64(L0)), _dp(_f64(ref0).reshape(self.nx)), _dp(x0))
        return x0

    def final_index(self, seed, logw):
        return int(lib().oc_final_index(self._h, seed, _dp(_f64(logw))))

    def sweep(self, seed, ref, A, LS, LSinv, cS, m0, L0, traces=True, nsteps_limit=0):
        N, T, nx = self.N, self.T, self.nx
        ref = _f64(ref).reshape(T, nx)
        A = _f64(A).reshape(nx, self.M)
        if traces:
            traj = np.empty((T, nx))
            X = np.empty((T, N, nx))
            ANC = np.empty((T - 1, N), np.int32)
        else:
            traj = X = ANC = None
        lw = np.empty(N)
        rc = lib().oc_sweep(
            self._h, seed, _dp(ref), _dp(A), _dp(_f64(LS)), _dp(_f64(LSinv)), cS, _dp(_f64(m0)), _dp(_f64(L0)),
            _dp(traj), _dp(X), _ip(ANC), _dp(lw), nsteps_limit,
        )
        assert rc == 0
        return traj, X, ANC, lw
