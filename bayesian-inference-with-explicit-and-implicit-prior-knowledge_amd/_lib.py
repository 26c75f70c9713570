"""ctypes binding of libpgas_hip.so (C ABI: include/pgas_hip.h) and the Engine wrapper.

There is no CPU fallback: if the library is missing or a call fails, an exception is raised.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PGAS_HIP_LIB") or os.path.join(_HERE, "libpgas_hip.so")  # override: kernel experiments only

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)


class PgasError(RuntimeError):
    pass


class _ModelDesc(C.Structure):
    _fields_ = [
        ("N", C.c_int32), ("T", C.c_int32), ("nx", C.c_int32), ("ny", C.c_int32), ("nu", C.c_int32),
        ("M", C.c_int32), ("D", C.c_int32),
        ("idx", _ip), ("sel", _ip), ("alpha", _dp), ("beta", _dp), ("nrm", C.c_double),
        ("H", _dp), ("LRinv", _dp), ("cR", C.c_double),
        ("m0", _dp), ("L0", _dp), ("y", _dp), ("u", _dp),
        ("device", C.c_int32), ("keep_logw_trace", C.c_int32), ("no_fast_variant", C.c_int32),
    ]


EXPORTS = [
    "pgas_create", "pgas_destroy", "pgas_last_error", "pgas_segment_size", "pgas_set_params", "pgas_set_params_dev", "pgas_get_params", "pgas_basis_eval",
    "pgas_aux_states", "pgas_init_state", "pgas_step", "pgas_sweep", "pgas_get_traces", "pgas_last_final_index",
    "pgas_suffstats", "pgas_set_profiling", "pgas_get_profile", "pgas_set_option",
    "pgas_systematic_resample", "pgas_reconstruct_trajectory",
    "pgas_trace_layout", "pgas_trace_row",
    "pgas_shard_setup", "pgas_shard_buffers", "pgas_shard_layout", "pgas_shard_block", "pgas_shard_set_peer_block", "pgas_shard_run", "pgas_ipc_export", "pgas_ipc_open",
    "pgas_hip_runtime_version", "pgas_shard_unique_id", "pgas_shard_comm_init", "pgas_shard_sweep", "pgas_shard_set_collective", "pgas_get_launch_info", "pgas_shard_probe_collective", "pgas_detmath_eval",
    "pgas_m_rng_uniform", "pgas_m_rng_normal", "pgas_m_rng_student_t", "pgas_m_rng_student_t_host", "pgas_m_rng_chi2", "pgas_m_set_time_source", "pgas_m_rng_uniform_dev", "pgas_systematic_resample_dev", "pgas_m_mniw_solve", "pgas_m_mniw_trisolve", "pgas_m_check", "pgas_m_stats_gather_update", "pgas_m_weighted_stats",
    "pgas_m_mniw_solve_n", "pgas_m_mniw_trisolve_n", "pgas_m_stats_gather_update_n", "pgas_m_weighted_stats_n", "pgas_m_expr_eval",
    "pgas_m_rng_student_t_df", "pgas_m_mniw_draw", "pgas_m_hilbert_basis", "pgas_m_lbm_diff",
]

ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int32, C.c_void_p)   # pgas_allgather_fn (include/pgas_hip.h)

_lib = None


def load():
    """Load libpgas_hip.so; raises PgasError if it has not been built (python __graft_entry__.py)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PgasError(f"{LIB_PATH} not found: build it with `python __graft_entry__.py` (hipcc, gfx950); there is no CPU fallback")
    L = C.CDLL(LIB_PATH)

    class _Missing:   # an older build (tools/ab_bench.py variants) may lack newer entry points: they fail when called, not at load time
        restype = argtypes = None

        def __init__(self, name):
            self.name = name

        def __call__(self, *a):
            raise PgasError(f"{LIB_PATH} does not export {self.name}")

    for name in EXPORTS:
        if not hasattr(L, name):
            setattr(L, name, _Missing(name))
    vp, u64, i32, i64 = C.c_void_p, C.c_uint64, C.c_int32, C.c_int64
    L.pgas_create.restype = C.c_int
    L.pgas_create.argtypes = [C.POINTER(_ModelDesc), C.POINTER(vp)]
    L.pgas_destroy.restype = None
    L.pgas_destroy.argtypes = [vp]
    L.pgas_last_error.restype = C.c_char_p
    L.pgas_last_error.argtypes = [vp]
    L.pgas_segment_size.restype = i32
    L.pgas_set_params.restype = C.c_int
    L.pgas_set_params.argtypes = [vp, vp, _dp, _dp, C.c_double, vp]
    L.pgas_set_params_dev.restype = C.c_int
    L.pgas_set_params_dev.argtypes = [vp, vp, vp, vp]
    L.pgas_get_params.restype = C.c_int
    L.pgas_get_params.argtypes = [vp, _dp, _dp, _dp, vp]
    L.pgas_basis_eval.restype = C.c_int
    L.pgas_basis_eval.argtypes = [vp, vp, i64, i32, vp, vp]
    L.pgas_aux_states.restype = C.c_int
    L.pgas_aux_states.argtypes = [vp, vp, i32, vp, vp]
    L.pgas_init_state.restype = C.c_int
    L.pgas_init_state.argtypes = [vp, u64, _dp, vp, vp]
    L.pgas_step.restype = C.c_int
    L.pgas_step.argtypes = [vp, i32, u64, vp, vp, _dp, vp, vp, vp, vp]
    L.pgas_sweep.restype = C.c_int
    L.pgas_sweep.argtypes = [vp, u64, vp, vp, vp]
    L.pgas_get_traces.restype = C.c_int
    L.pgas_get_traces.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]
    L.pgas_last_final_index.restype = C.c_int
    L.pgas_last_final_index.argtypes = [vp, C.POINTER(i64), vp]
    L.pgas_suffstats.restype = C.c_int
    L.pgas_suffstats.argtypes = [vp, vp, vp, vp, vp, vp]
    L.pgas_set_profiling.restype = C.c_int
    L.pgas_set_profiling.argtypes = [vp, i32]
    L.pgas_get_profile.restype = C.c_int
    L.pgas_get_profile.argtypes = [vp, C.POINTER(i64), C.POINTER(C.c_double), C.POINTER(i64), C.POINTER(C.c_double), vp]
    L.pgas_set_option.restype = C.c_int
    L.pgas_set_option.argtypes = [vp, i32, i64]
    L.pgas_systematic_resample.restype = C.c_int
    L.pgas_systematic_resample.argtypes = [vp, C.c_double, vp, vp, vp]
    L.pgas_reconstruct_trajectory.restype = C.c_int
    L.pgas_reconstruct_trajectory.argtypes = [vp, vp, vp, i32, i32, i64, vp, vp]
    L.pgas_shard_setup.restype = C.c_int
    L.pgas_shard_setup.argtypes = [vp, i32, i32]
    L.pgas_shard_buffers.restype = C.c_int
    L.pgas_shard_buffers.argtypes = [vp, C.POINTER(vp), C.POINTER(i64)]
    L.pgas_shard_layout.restype = C.c_int
    L.pgas_shard_layout.argtypes = [vp, i32, C.POINTER(i64)]
    L.pgas_shard_block.restype = C.c_int
    L.pgas_shard_block.argtypes = [vp, i32, i32, C.POINTER(vp), C.POINTER(i64)]
    L.pgas_shard_set_peer_block.restype = C.c_int
    L.pgas_shard_set_peer_block.argtypes = [vp, i32, i32, i32, vp]
    L.pgas_trace_layout.restype = C.c_int
    L.pgas_trace_layout.argtypes = [vp, i32, C.POINTER(i64)]
    L.pgas_trace_row.restype = C.c_int
    L.pgas_trace_row.argtypes = [vp, i32, i32, C.POINTER(vp)]
    L.pgas_shard_run.restype = C.c_int
    L.pgas_shard_run.argtypes = [vp, i32, i32, i32, u64, vp, vp, vp]
    L.pgas_shard_unique_id.restype = C.c_int
    L.pgas_shard_unique_id.argtypes = [C.c_char_p]
    L.pgas_shard_comm_init.restype = C.c_int
    L.pgas_shard_comm_init.argtypes = [vp, C.c_char_p]
    L.pgas_shard_sweep.restype = C.c_int
    L.pgas_shard_sweep.argtypes = [vp, u64, vp, vp, i32, vp]
    L.pgas_shard_set_collective.restype = C.c_int
    L.pgas_shard_set_collective.argtypes = [vp, ALLGATHER_FN, vp]
    L.pgas_detmath_eval.restype = C.c_int
    L.pgas_detmath_eval.argtypes = [i32, i32, vp, vp, vp, i64, vp, vp, vp, vp]
    L.pgas_shard_probe_collective.restype = C.c_int
    L.pgas_shard_probe_collective.argtypes = [vp, i32, vp]
    L.pgas_get_launch_info.restype = C.c_int
    L.pgas_get_launch_info.argtypes = [vp, C.POINTER(i32)]
    L.pgas_ipc_export.restype = C.c_int
    L.pgas_ipc_export.argtypes = [vp, i32, i32, C.c_char_p]
    L.pgas_hip_runtime_version.restype = C.c_int32
    L.pgas_hip_runtime_version.argtypes = []
    L.pgas_ipc_open.restype = C.c_int
    L.pgas_ipc_open.argtypes = [vp, C.c_char_p, C.POINTER(vp)]
    u32 = C.c_uint32
    L.pgas_m_rng_uniform.restype = C.c_double
    L.pgas_m_rng_uniform.argtypes = [u64, u32, u32]
    L.pgas_m_rng_normal.restype = C.c_int
    L.pgas_m_rng_normal.argtypes = [vp, u64, u32, u32, i64, i64, i32, vp, vp]
    L.pgas_m_rng_student_t.restype = C.c_int
    L.pgas_m_rng_student_t.argtypes = [vp, u64, u32, u32, i64, i64, vp, vp, vp]
    L.pgas_m_rng_student_t_host.restype = C.c_int
    L.pgas_m_rng_student_t_host.argtypes = [u64, u32, u32, i64, i64, _dp, _dp]
    L.pgas_m_set_time_source.restype = C.c_int
    L.pgas_m_set_time_source.argtypes = [vp, vp]
    L.pgas_m_rng_uniform_dev.restype = C.c_int
    L.pgas_m_rng_uniform_dev.argtypes = [vp, u64, u32, u32, vp, vp]
    L.pgas_systematic_resample_dev.restype = C.c_int
    L.pgas_systematic_resample_dev.argtypes = [vp, vp, vp, vp, vp]
    L.pgas_m_rng_chi2.restype = C.c_int
    L.pgas_m_rng_chi2.argtypes = [vp, u64, u32, u32, i64, i64, vp, vp, vp]
    L.pgas_m_mniw_solve.restype = C.c_int
    L.pgas_m_mniw_solve.argtypes = [vp, i64, i32, C.c_double, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    L.pgas_m_mniw_trisolve.restype = C.c_int
    L.pgas_m_mniw_trisolve.argtypes = [vp, i64, i32, vp, vp, vp, vp, vp, vp]
    L.pgas_m_rng_student_t_df.restype = C.c_int
    L.pgas_m_rng_student_t_df.argtypes = [vp, u64, C.c_uint32, C.c_uint32, i64, i64, vp, vp, C.c_double, C.c_double, vp, vp]
    L.pgas_m_mniw_draw.restype = C.c_int
    L.pgas_m_mniw_draw.argtypes = [vp, i64, C.c_double, vp, vp, vp, vp, vp, vp, C.c_double, C.c_double, vp, vp, vp]
    L.pgas_m_lbm_diff.restype = C.c_int
    L.pgas_m_lbm_diff.argtypes = [vp, i64, i32, vp, vp, vp, vp, vp, vp, C.c_double, C.c_double, vp, vp, vp, vp]
    L.pgas_m_hilbert_basis.restype = C.c_int
    L.pgas_m_hilbert_basis.argtypes = [vp, i64, i32, i32, vp, i32, vp, i32, C.POINTER(i32), _dp, _dp, _dp, _dp, vp, vp, vp]
    L.pgas_m_expr_eval.restype = C.c_int
    L.pgas_m_expr_eval.argtypes = [vp, i64, vp, i32, vp, i32, i32, i32, C.POINTER(i32), i32, vp, i32, vp, vp, i32, C.POINTER(vp), C.POINTER(i32), i32, i32,
                                   vp, vp, C.c_double, vp, vp]
    L.pgas_m_mniw_solve_n.restype = C.c_int
    L.pgas_m_mniw_solve_n.argtypes = [vp, i64, i32, i32, C.c_double, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    L.pgas_m_mniw_trisolve_n.restype = C.c_int
    L.pgas_m_mniw_trisolve_n.argtypes = [vp, i64, i32, i32, vp, vp, vp, vp, vp, vp]
    L.pgas_m_weighted_stats_n.restype = C.c_int
    L.pgas_m_weighted_stats_n.argtypes = [vp, i64, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    L.pgas_m_stats_gather_update_n.restype = C.c_int
    L.pgas_m_stats_gather_update_n.argtypes = [vp, i64, i32, i32, C.c_double, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    L.pgas_m_weighted_stats.restype = C.c_int
    L.pgas_m_weighted_stats.argtypes = [vp, i64, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    L.pgas_m_check.restype = C.c_int
    L.pgas_m_check.argtypes = [vp, vp]
    L.pgas_m_stats_gather_update.restype = C.c_int
    L.pgas_m_stats_gather_update.argtypes = [vp, i64, i32, C.c_double, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    _lib = L
    return L


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _hp(a):
    return a.ctypes.data_as(_dp)


def detmath_eval(which, x=None, y=None, words=None, device=None):
    """Test hook (pgas_detmath_eval): the shared arithmetic primitives evaluated on the device; returns numpy arrays."""
    L = load()
    dev = torch.device("cuda", torch.cuda.current_device() if device is None else torch.device(device).index or 0)
    f = lambda a: None if a is None else torch.as_tensor(np.ascontiguousarray(a, dtype=np.float64), device=dev)  # noqa: E731
    xd, yd = f(x), f(y)
    wd = None if words is None else torch.as_tensor(np.ascontiguousarray(words, dtype=np.uint32).view(np.int32), device=dev)
    n = int(xd.numel()) if xd is not None else int(wd.numel() // 6)
    o0 = torch.empty(n, dtype=torch.float64, device=dev)
    o1 = torch.empty(n, dtype=torch.float64, device=dev)
    ow = torch.empty(4 * n, dtype=torch.int32, device=dev)
    p = lambda t: 0 if t is None else t.data_ptr()  # noqa: E731
    rc = L.pgas_detmath_eval(dev.index, int(which), p(xd), p(yd), p(wd), n, o0.data_ptr(), o1.data_ptr(), ow.data_ptr(),
                             C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
    if rc != 0:
        raise PgasError(f"pgas_detmath_eval failed ({rc})")
    torch.cuda.synchronize(dev)
    return o0.cpu().numpy(), o1.cpu().numpy(), ow.cpu().numpy().view(np.uint32).reshape(n, 4)


def student_t_host(seed, stream, t, nu, p0=0):
    """Student-t variates of counters (particle p0 .. p0 + len(nu) - 1, t, stream) computed on the HOST by the library's own arithmetic:
    bit-identical to what k_rng_student_t produces on the device (pgas_m_rng_student_t_host)."""
    nu = np.ascontiguousarray(np.atleast_1d(nu), dtype=np.float64)
    out = np.empty_like(nu)
    rc = load().pgas_m_rng_student_t_host(int(seed), int(stream), int(t), int(p0), nu.size, _hp(nu), _hp(out))
    if rc != 0:
        raise PgasError(f"pgas_m_rng_student_t_host failed ({rc})")
    return out


class _DevView:
    """Minimal __cuda_array_interface__ carrier so torch can wrap library-owned device memory."""

    def __init__(self, ptr, shape, typestr, owner):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False), "version": 2}
        self._owner = owner


class Engine:
    """One pgas_ctx: model tables on the device + launch methods on torch tensors."""

    def __init__(self, N, observations, inputs, init_state_mean, init_state_cov, likelihood, basis_map, device=None, keep_logw_trace=False):
        if not torch.cuda.is_available():
            raise PgasError("no HIP device visible: the engine has no CPU path")
        self.lib = load()
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else torch.device(device).index or 0)
        y = _f64(observations)
        self.T = y.shape[0]
        y = y.reshape(self.T, -1)
        u = _f64(inputs).reshape(self.T, -1) if inputs is not None and np.size(inputs) else np.zeros((self.T, 0))
        m0 = _f64(init_state_mean).reshape(-1)
        self.N, self.nx, self.ny, self.nu = int(N), m0.shape[0], y.shape[1], u.shape[1]
        if likelihood.nx != self.nx or likelihood.ny != self.ny:
            raise ValueError("likelihood dimensions do not match the state / observation dimensions")
        b = basis_map.basis
        self.M, self.D = b.M, b.D
        self._keep = dict(
            idx=np.ascontiguousarray(b.indices, dtype=np.int32), sel=np.ascontiguousarray(basis_map.sel, dtype=np.int32),
            alpha=_f64(basis_map.alpha), beta=_f64(basis_map.beta), H=_f64(likelihood.H), LRinv=_f64(likelihood.LRinv),
            m0=m0, L0=_f64(np.linalg.cholesky(np.atleast_2d(_f64(init_state_cov)))), y=y, u=u if self.nu else np.zeros(1),
        )
        k = self._keep
        d = _ModelDesc(
            self.N, self.T, self.nx, self.ny, self.nu, self.M, self.D,
            k["idx"].ctypes.data_as(_ip), k["sel"].ctypes.data_as(_ip), _hp(k["alpha"]), _hp(k["beta"]), b.norm,
            _hp(k["H"]), _hp(k["LRinv"]), likelihood.cR, _hp(k["m0"]), _hp(k["L0"]), _hp(k["y"]),
            _hp(k["u"]) if self.nu else None, self.device.index, 1 if keep_logw_trace else 0,
            1 if os.environ.get("PGAS_NO_FAST_VARIANT", "0") == "1" else 0,
        )
        h = C.c_void_p()
        rc = self.lib.pgas_create(C.byref(d), C.byref(h))
        if rc != 0:
            raise PgasError(f"pgas_create failed ({rc}): {self.lib.pgas_last_error(None).decode()}")
        self._h = h
        self.keep_logw_trace = bool(keep_logw_trace)

    def close(self):
        if getattr(self, "_h", None):
            self.lib.pgas_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -------------------------------------------------------------- helpers
    def _chk(self, rc, what):
        if rc != 0:
            raise PgasError(f"{what} failed ({rc}): {self.lib.pgas_last_error(self._h).decode()}")

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _dev(self, t, dtype=torch.float64, shape=None):
        if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == dtype and t.is_contiguous()):
            t = torch.as_tensor(np.asarray(t) if not isinstance(t, torch.Tensor) else t, dtype=dtype, device=self.device).contiguous()
        if shape is not None and tuple(t.shape) != tuple(shape):
            t = t.reshape(shape).contiguous()
        return t

    # -------------------------------------------------------------- calls
    def set_params(self, coeff_mat, error_cov):
        """coeff_mat (nx,M) tensor/array, error_cov (nx,nx).  An error_cov that already lives on the engine's device is factored THERE
        (pgas_set_params_dev: no synchronisation, no host round trip -- the Gibbs loop's case) and None is returned; a host array is
        factored with NumPy and (LS, LSinv, cS) returned.  The two factorisations agree to rounding, not bit for bit: get_params() reads
        back what the kernels use."""
        A = self._dev(coeff_mat, shape=(self.nx, self.M))
        if isinstance(error_cov, torch.Tensor) and error_cov.is_cuda and error_cov.device == self.device and os.environ.get("PGAS_HOST_PARAMS", "0") != "1":
            Sd = error_cov.detach().to(torch.float64).reshape(self.nx, self.nx).contiguous()
            self._A, self._S = A, Sd   # keep alive until the pack kernel has run
            self._chk(self.lib.pgas_set_params_dev(self._h, A.data_ptr(), Sd.data_ptr(), self._stream()), "pgas_set_params_dev")
            return None
        S = np.atleast_2d(np.asarray(error_cov.detach().cpu() if isinstance(error_cov, torch.Tensor) else error_cov, dtype=np.float64))
        LS = _f64(np.linalg.cholesky(S))
        LSinv = _f64(np.linalg.inv(LS))
        cS = float(-0.5 * self.nx * np.log(2 * np.pi) - np.sum(np.log(np.diag(LS))))
        self._A = A  # keep alive until the pack kernel has run
        self._chk(self.lib.pgas_set_params(self._h, A.data_ptr(), _hp(LS), _hp(LSinv), cS, self._stream()), "pgas_set_params")
        return LS, LSinv, cS

    def get_params(self):
        """(LS, LSinv, cS) the kernels currently use (synchronises): Cholesky factor of error_cov, its inverse, the normalising constant."""
        LS, LSinv, cS = np.zeros((self.nx, self.nx)), np.zeros((self.nx, self.nx)), C.c_double()
        self._chk(self.lib.pgas_get_params(self._h, _hp(LS), _hp(LSinv), C.byref(cS), self._stream()), "pgas_get_params")
        return LS, LSinv, float(cS.value)

    def basis_eval(self, x, t):
        x = self._dev(x).reshape(-1, self.nx)
        phi = torch.empty((x.shape[0], self.M), dtype=torch.float64, device=self.device)
        self._chk(self.lib.pgas_basis_eval(self._h, x.data_ptr(), x.shape[0], t, phi.data_ptr(), self._stream()), "pgas_basis_eval")
        return phi

    def aux_states(self, x, t):
        x = self._dev(x, shape=(self.N, self.nx))
        aux = torch.empty_like(x)
        self._chk(self.lib.pgas_aux_states(self._h, x.data_ptr(), t, aux.data_ptr(), self._stream()), "pgas_aux_states")
        return aux

    def init_state(self, seed, ref0):
        x0 = torch.empty((self.N, self.nx), dtype=torch.float64, device=self.device)
        r = _f64(ref0).reshape(self.nx)
        self._chk(self.lib.pgas_init_state(self._h, seed, _hp(r), x0.data_ptr(), self._stream()), "pgas_init_state")
        torch.cuda.current_stream(self.device).synchronize()  # r is host memory read asynchronously
        return x0

    def step(self, t, seed, log_weights, state, ref_t):
        x = self._dev(state, shape=(self.N, self.nx))
        lw = None if log_weights is None else self._dev(log_weights, shape=(self.N,))
        r = _f64(ref_t.detach().cpu() if isinstance(ref_t, torch.Tensor) else ref_t).reshape(self.nx)
        lw_new = torch.empty(self.N, dtype=torch.float64, device=self.device)
        x_new = torch.empty((self.N, self.nx), dtype=torch.float64, device=self.device)
        anc = torch.empty(self.N, dtype=torch.int32, device=self.device)
        self._chk(
            self.lib.pgas_step(self._h, t, seed, None if lw is None else lw.data_ptr(), x.data_ptr(), _hp(r), lw_new.data_ptr(),
                               x_new.data_ptr(), anc.data_ptr(), self._stream()),
            "pgas_step",
        )
        torch.cuda.current_stream(self.device).synchronize()
        return lw_new, x_new, anc

    def sweep(self, seed, ref_state):
        ref = self._dev(ref_state, shape=(self.T, self.nx))
        traj = torch.empty((self.T, self.nx), dtype=torch.float64, device=self.device)
        self._chk(self.lib.pgas_sweep(self._h, seed, ref.data_ptr(), traj.data_ptr(), self._stream()), "pgas_sweep")
        self._ref_keepalive = ref
        return traj

    TRACE_X, TRACE_LA, TRACE_H, TRACE_LN, TRACE_ANC = range(5)

    def trace_layout(self, kind):
        """(rows, rows per block, blocks, bytes per row) of trace `kind` (pgas_trace_layout)."""
        v = (C.c_int64 * 4)()
        self._chk(self.lib.pgas_trace_layout(self._h, int(kind), v), "pgas_trace_layout")
        return tuple(int(x) for x in v)

    def trace_row(self, kind, t):
        """Device pointer of time row t of trace `kind` (pgas_trace_row)."""
        p = C.c_void_p()
        self._chk(self.lib.pgas_trace_row(self._h, int(kind), int(t), C.byref(p)), "pgas_trace_row")
        return int(p.value)

    def traces_blocks(self, kind, row_shape, dtype):
        """The row blocks of trace `kind` as torch views of library-owned memory, in time order."""
        rows, rpb, nblk, _ = self.trace_layout(kind)
        return [self.dev_tensor(self.trace_row(kind, b * rpb), (min(rpb, rows - b * rpb),) + tuple(row_shape), dtype) for b in range(nblk)]

    def traces(self, copy_blocks=True):
        """(state_trace (T,N,nx), ancestor_trace (T-1,N) int32, logw_last (N), logw_trace (T,N)|None).  Views of library-owned memory on
        an unsharded context; a context that keeps its traces in row blocks (shards, PGAS_OPT_TRACE_BLOCK_BYTES) returns concatenated
        COPIES of x / ancestors (None with copy_blocks=False: use traces_blocks for views of the blocks)."""
        px, pa, pl, pt = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
        if self.trace_layout(self.TRACE_X)[2] == 1 and self.trace_layout(self.TRACE_ANC)[2] == 1:
            px.value, pa.value = self.trace_row(self.TRACE_X, 0), self.trace_row(self.TRACE_ANC, 0)
            X = torch.as_tensor(_DevView(px.value, (self.T, self.N, self.nx), "<f8", self), device=self.device)
            A = torch.as_tensor(_DevView(pa.value, (max(self.T - 1, 1), self.N), "<i4", self), device=self.device)
        elif copy_blocks:
            X = torch.cat(self.traces_blocks(self.TRACE_X, (self.N, self.nx), torch.float64))
            A = torch.cat(self.traces_blocks(self.TRACE_ANC, (self.N,), torch.int32))
        else:
            X = A = None
        self._chk(self.lib.pgas_get_traces(self._h, None, None, C.byref(pl), C.byref(pt)), "pgas_get_traces")
        Lw = torch.as_tensor(_DevView(pl.value, (self.N,), "<f8", self), device=self.device)
        Lt = torch.as_tensor(_DevView(pt.value, (self.T, self.N), "<f8", self), device=self.device) if pt.value else None
        return X, A, Lw, Lt

    def last_final_index(self):
        v = C.c_int64()
        self._chk(self.lib.pgas_last_final_index(self._h, C.byref(v), self._stream()), "pgas_last_final_index")
        return int(v.value)

    def set_profiling(self, on):
        """False/0 = off, True/1 = on at the default sampling stride, n > 1 = time every n-th launch."""
        self._chk(self.lib.pgas_set_profiling(self._h, int(on)), "pgas_set_profiling")

    def profile(self):
        """(k_step launches, their total ms, k_propagate launches, their total ms) of the last sweep (synchronises)."""
        n, ms, pn, pm = C.c_int64(), C.c_double(), C.c_int64(), C.c_double()
        self._chk(self.lib.pgas_get_profile(self._h, C.byref(n), C.byref(ms), C.byref(pn), C.byref(pm), self._stream()), "pgas_get_profile")
        return int(n.value), float(ms.value), int(pn.value), float(pm.value)

    def set_option(self, option, value):
        self._chk(self.lib.pgas_set_option(self._h, int(option), int(value)), "pgas_set_option")

    # -------------------------------------------------------------- src/Filtering.py free functions
    _utility = {}

    @classmethod
    def utility(cls, N, device=None):
        """A context that only serves pgas_systematic_resample / pgas_reconstruct_trajectory for N particles (trivial model)."""
        from .descriptors import BasisMap, GaussianLikelihood, HilbertBasis

        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        key = (dev.index or 0, int(N))
        if key not in cls._utility:
            basis = HilbertBasis(np.array([[1]], dtype=np.int32), np.array([[-1.0, 1.0]]))
            cls._utility[key] = cls(int(N), np.zeros((1, 1)), None, np.zeros(1), np.eye(1), GaussianLikelihood(np.eye(1), np.eye(1)),
                                    BasisMap(basis, [0]), device=dev)
        return cls._utility[key]

    def systematic_resample(self, u, logw):
        """u: a float, or a device tensor holding the uniform (read at execution time: graph-captured steps)."""
        lw = self._dev(logw, shape=(self.N,))
        idx = torch.empty(self.N, dtype=torch.int32, device=self.device)
        if isinstance(u, torch.Tensor):
            self._chk(self.lib.pgas_systematic_resample_dev(self._h, u.data_ptr(), lw.data_ptr(), idx.data_ptr(), self._stream()), "pgas_systematic_resample_dev")
        else:
            self._chk(self.lib.pgas_systematic_resample(self._h, float(u), lw.data_ptr(), idx.data_ptr(), self._stream()), "pgas_systematic_resample")
        return idx

    def reconstruct_trajectory(self, particles, ancestry, idx):
        P = particles if isinstance(particles, torch.Tensor) else torch.as_tensor(np.asarray(particles, dtype=np.float64))
        P = P.to(device=self.device, dtype=torch.float64)
        if P.dim() == 2:
            P = P.unsqueeze(-1)
        P = P.contiguous()
        T, N, nx = P.shape
        if N != self.N:
            raise ValueError(f"particle count {N} does not match the context ({self.N})")
        A = ancestry if isinstance(ancestry, torch.Tensor) else torch.as_tensor(np.asarray(ancestry))
        A = A.to(device=self.device, dtype=torch.int32).contiguous()   # the reference stores float64 indices (Q2)
        traj = torch.empty((T, nx), dtype=torch.float64, device=self.device)
        self._chk(self.lib.pgas_reconstruct_trajectory(self._h, P.data_ptr(), A.data_ptr(), T, nx, int(idx), traj.data_ptr(), self._stream()),
                  "pgas_reconstruct_trajectory")
        return traj

    # -------------------------------------------------------------- particle sharding (pgas_amd/sharded.py)
    def shard_setup(self, rank, world):
        self._chk(self.lib.pgas_shard_setup(self._h, rank, world), "pgas_shard_setup")
        self.rank, self.world = rank, world

    def shard_buffers(self):
        """(list of 17 device pointers, (nsegp, N_local, T)) -- see include/pgas_hip.h."""
        out = (C.c_void_p * 17)()
        sz = (C.c_int64 * 3)()
        self._chk(self.lib.pgas_shard_buffers(self._h, out, sz), "pgas_shard_buffers")
        return [int(p or 0) for p in out], tuple(int(v) for v in sz)

    def shard_layout(self, which):
        """(blocks, rows per block) of peer-visible buffer `which` (0, 1: segment cumsums; 2..6: la, h, ln, x, anc)."""
        v = (C.c_int64 * 2)()
        self._chk(self.lib.pgas_shard_layout(self._h, int(which), v), "pgas_shard_layout")
        return int(v[0]), int(v[1])

    def shard_block(self, which, blk):
        """(device pointer, bytes) of block `blk` of this rank's buffer `which`."""
        p, n = C.c_void_p(), C.c_int64()
        self._chk(self.lib.pgas_shard_block(self._h, int(which), int(blk), C.byref(p), C.byref(n)), "pgas_shard_block")
        return int(p.value), int(n.value)

    def shard_set_peer_block(self, peer, which, blk, ptr):
        self._chk(self.lib.pgas_shard_set_peer_block(self._h, int(peer), int(which), int(blk), C.c_void_p(int(ptr))), "pgas_shard_set_peer_block")

    def shard_run(self, phase, t=0, t_aux=0, seed=0, ref=None, traj=None):
        self._chk(
            self.lib.pgas_shard_run(self._h, phase, t, t_aux, seed, None if ref is None else ref.data_ptr(),
                                    None if traj is None else traj.data_ptr(), self._stream()),
            "pgas_shard_run",
        )

    def shard_unique_id(self):
        buf = C.create_string_buffer(128)
        rc = self.lib.pgas_shard_unique_id(buf)
        if rc:
            raise PgasError("pgas_shard_unique_id failed (librccl.so not loadable?)")
        return bytes(buf.raw)

    def shard_comm_init(self, id128):
        self._chk(self.lib.pgas_shard_comm_init(self._h, id128), "pgas_shard_comm_init")

    def shard_set_collective(self, fn):
        """Install a host all-gather `fn(parity, stream_ptr) -> int` in place of RCCL for pgas_shard_sweep (tests); None removes it."""
        if fn is None:
            self._ag_cb = None
            self._chk(self.lib.pgas_shard_set_collective(self._h, C.cast(None, ALLGATHER_FN), None), "pgas_shard_set_collective")
            return

        def tramp(user, parity, stream):
            try:
                return int(fn(int(parity), stream) or 0)
            except Exception as exc:  # an exception must not unwind through the C frames
                import traceback

                self._ag_error = "".join(traceback.format_exception(type(exc), exc, exc.__traceback__))
                return -1

        self._ag_cb = ALLGATHER_FN(tramp)   # keep the trampoline alive as long as the library may call it
        self._chk(self.lib.pgas_shard_set_collective(self._h, self._ag_cb, None), "pgas_shard_set_collective")

    def shard_probe_collective(self, reps=200):
        """Mean duration (us) of the step's RCCL all-gather issued `reps` times back to back on an idle stream; None without RCCL."""
        if getattr(self, "_ag_cb", None) is not None:
            return None
        torch.cuda.synchronize(self.device)
        self._chk(self.lib.pgas_shard_probe_collective(self._h, 3, self._stream()), "pgas_shard_probe_collective")   # warm-up
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        self._chk(self.lib.pgas_shard_probe_collective(self._h, int(reps), self._stream()), "pgas_shard_probe_collective")
        e1.record()
        torch.cuda.synchronize(self.device)
        return e0.elapsed_time(e1) * 1e3 / reps

    def launch_info(self):
        """dict(chunk, local_groups, JP, P) of the last sweep (pgas_get_launch_info)."""
        v = (C.c_int32 * 4)()
        self._chk(self.lib.pgas_get_launch_info(self._h, v), "pgas_get_launch_info")
        g = int(v[1]) & 15
        return dict(chunk=int(v[0]), local_groups=g == 1, groups={0: "k_groups", 1: "local", 2: "tail", 3: "small"}[g], small=g == 3, graph=bool(int(v[1]) & 16), mfma=bool(int(v[1]) & 32), JP=int(v[2]), P=int(v[3]))

    def shard_sweep(self, seed, ref, traj, propagate_chunk=0):
        self._ag_error = None
        rc = self.lib.pgas_shard_sweep(self._h, int(seed), ref.data_ptr(), traj.data_ptr(), int(propagate_chunk), self._stream())
        if rc != 0 and getattr(self, "_ag_error", None):
            raise PgasError(f"pgas_shard_sweep: the all-gather callback raised:\n{self._ag_error}")
        self._chk(rc, "pgas_shard_sweep")

    def ipc_export(self, which, blk=0):
        buf = C.create_string_buffer(64)
        self._chk(self.lib.pgas_ipc_export(self._h, int(which), int(blk), buf), "pgas_ipc_export")
        return bytes(buf.raw)

    def ipc_open(self, handle):
        p = C.c_void_p()
        self._chk(self.lib.pgas_ipc_open(self._h, handle, C.byref(p)), "pgas_ipc_open")
        return int(p.value)

    def dev_tensor(self, ptr, shape, dtype):
        """torch view of library-owned device memory."""
        typestr = {torch.float64: "<f8", torch.int64: "<i8", torch.int32: "<i4"}[dtype]
        return torch.as_tensor(_DevView(ptr, shape, typestr, self), device=self.device)

    def rng_normal(self, seed, stream, t, n, ncol=1):
        """(n, ncol) standard normals of counters (particle 0..n-1, t, stream) -- Philox + Box-Muller on the device (k_rng_normal)."""
        out = torch.empty((int(n), int(ncol)), dtype=torch.float64, device=self.device)
        self._chk(self.lib.pgas_m_rng_normal(self._h, int(seed), int(stream), int(t), 0, int(n), int(ncol), out.data_ptr(), self._stream()), "pgas_m_rng_normal")
        return out

    def rng_chi2(self, seed, stream, t, nu):
        """chi^2(nu_p) variates for a device vector nu (k_rng_chi2)."""
        nu = nu.to(device=self.device, dtype=torch.float64).contiguous()
        out = torch.empty_like(nu)
        self._chk(self.lib.pgas_m_rng_chi2(self._h, int(seed), int(stream), int(t), 0, nu.numel(), nu.data_ptr(), out.data_ptr(), self._stream()), "pgas_m_rng_chi2")
        return out

    def suffstats(self, traj):
        traj = self._dev(traj, shape=(self.T, self.nx))
        T0 = torch.empty((self.M, self.nx), dtype=torch.float64, device=self.device)
        T1 = torch.empty((self.M, self.M), dtype=torch.float64, device=self.device)
        T2 = torch.empty((self.nx, self.nx), dtype=torch.float64, device=self.device)
        self._chk(self.lib.pgas_suffstats(self._h, traj.data_ptr(), T0.data_ptr(), T1.data_ptr(), T2.data_ptr(), self._stream()), "pgas_suffstats")
        return T0, T1, T2, float(self.T - 1)


class MarginalOps:
    """Device primitives of the marginalised family (include/pgas_marginal.h) for N particles on one device: Philox normals /
    uniforms / Student-t variates, batched MNIW solves, statistics gather + update, systematic resampling.  Hand-written HIP
    through the C ABI; there is no CPU path."""

    def __init__(self, N, device=None):
        self.eng = Engine.utility(int(N), device)
        self.N, self.device, self.lib = int(N), self.eng.device, self.eng.lib

    def _ptr(self, t):
        return 0 if t is None else t.data_ptr()

    def _vec(self, n=None, cols=None):
        shape = (self.N if n is None else n,) if cols is None else (self.N if n is None else n, cols)
        return torch.empty(shape, dtype=torch.float64, device=self.device)

    def uniform(self, seed, stream, t):
        return float(self.lib.pgas_m_rng_uniform(int(seed), int(stream), int(t)))

    def set_time_source(self, t_dev):
        """t_dev: a uint32-sized device tensor (int32 works) the random-number kernels read their time index from, or None."""
        self._t_dev = t_dev   # keep it alive
        self.eng._chk(self.lib.pgas_m_set_time_source(self.eng._h, 0 if t_dev is None else t_dev.data_ptr()), "pgas_m_set_time_source")

    def uniform_dev(self, seed, stream, t):
        """The same uniform as `uniform`, produced on the device into a one-element tensor (time index from the time source if set)."""
        out = torch.empty(1, dtype=torch.float64, device=self.device)
        self.eng._chk(self.lib.pgas_m_rng_uniform_dev(self.eng._h, int(seed), int(stream), int(t), out.data_ptr(), self.eng._stream()), "pgas_m_rng_uniform_dev")
        return out

    def normal(self, seed, stream, t, ncol):
        out = self._vec(cols=int(ncol))
        self.eng._chk(self.lib.pgas_m_rng_normal(self.eng._h, int(seed), int(stream), int(t), 0, self.N, int(ncol), out.data_ptr(), self.eng._stream()),
                      "pgas_m_rng_normal")
        return out

    def student_t(self, seed, stream, t, nu):
        nu = nu.to(device=self.device, dtype=torch.float64).contiguous()
        out = self._vec()
        self.eng._chk(self.lib.pgas_m_rng_student_t(self.eng._h, int(seed), int(stream), int(t), 0, self.N, nu.data_ptr(), out.data_ptr(),
                                                    self.eng._stream()), "pgas_m_rng_student_t")
        return out

    def student_t_df(self, seed, stream, t, anc, src, nu0, nu_scale):
        """Student-t variates with nu[p] = nu0 + nu_scale * src[anc[p]] formed inside the kernel."""
        out = self._vec()
        a = anc.to(device=self.device, dtype=torch.int32).contiguous()
        self.eng._chk(self.lib.pgas_m_rng_student_t_df(self.eng._h, int(seed), int(stream), int(t), 0, self.N, a.data_ptr(), src.contiguous().data_ptr(),
                                                       float(nu0), float(nu_scale), out.data_ptr(), self.eng._stream()), "pgas_m_rng_student_t_df")
        return out

    def mniw_draw(self, scale, anc, m, c, q, T2, T3, P2, P3, t):
        """xi = m + sqrt((P2 + scale T2[a] - q[a]) / (P3 + scale T3[a])) t sqrt(c + 1) in one launch (pgas_m_mniw_draw)."""
        n = m.shape[0]
        out = self._vec(n)
        a = anc.to(device=self.device, dtype=torch.int32).contiguous()
        self.eng._chk(self.lib.pgas_m_mniw_draw(self.eng._h, n, float(scale), a.data_ptr(), m.contiguous().data_ptr(), c.contiguous().data_ptr(),
                                                q.contiguous().data_ptr(), T2.contiguous().data_ptr(), T3.contiguous().data_ptr(), float(P2), float(P3),
                                                t.contiguous().data_ptr(), out.data_ptr(), self.eng._stream()), "pgas_m_mniw_draw")
        return out

    def lbm_diff(self, M, T2, T3, sol1, sol2, P2, P3, r2, r3):
        """lbm(prior + T) - lbm(prior + T + R) per particle from the two solves' q / logdet (pgas_m_lbm_diff; n = 1)."""
        n = T2.shape[0]
        out = self._vec(n)
        r2, r3 = r2.reshape(-1).contiguous(), r3.reshape(-1).contiguous()
        self.eng._chk(self.lib.pgas_m_lbm_diff(self.eng._h, n, int(M), T2.contiguous().data_ptr(), T3.contiguous().data_ptr(), sol1["q"].data_ptr(),
                                               sol1["logdet"].data_ptr(), sol2["q"].data_ptr(), sol2["logdet"].data_ptr(), float(P2), float(P3),
                                               r2.data_ptr(), r3.data_ptr(), out.data_ptr(), self.eng._stream()), "pgas_m_lbm_diff")
        return out

    def hilbert_basis(self, bmap, state, input=None):
        """descriptors.BasisMap.batch for (n, n_x) states and one (n_u,) input in one launch (pgas_m_hilbert_basis) -> (n, M)."""
        b = bmap.basis
        n = state.shape[0]
        state = state.reshape(n, -1).contiguous()
        nx = state.shape[1]
        inp = None if input is None or input.numel() == 0 else input.reshape(-1).contiguous()
        nu = 0 if inp is None else inp.numel()
        cache = bmap.__dict__.setdefault("_idx_dev", {})
        if self.device not in cache:
            cache[self.device] = torch.as_tensor(np.ascontiguousarray(b.indices, dtype=np.int32), device=self.device)
        host = [np.ascontiguousarray(a, dtype=np.float64) for a in (bmap.div, b.center, b.L, b.size)]   # kept alive over the call
        sel = np.ascontiguousarray(bmap.sel, dtype=np.int32)
        out = torch.empty((n, b.M), dtype=torch.float64, device=self.device)
        self.eng._chk(self.lib.pgas_m_hilbert_basis(self.eng._h, n, b.M, b.D, state.data_ptr(), nx, self._ptr(inp), nu, sel.ctypes.data_as(C.POINTER(C.c_int32)),
                                                    *[a.ctypes.data_as(_dp) for a in host], cache[self.device].data_ptr(), out.data_ptr(),
                                                    self.eng._stream()), "pgas_m_hilbert_basis")
        return out

    def mniw_solve(self, P0, P1, T0, T1, scale=1.0, anc=None, R0=None, R1=None, phi=None, want=("m", "c", "q", "logdet"), keep_factor=False):
        """eta0 = P0 + scale T0[anc] (+R0), eta1 = P1 + scale T1[anc] (+R1) per particle -> dict of (n,) tensors (see pgas_m_mniw_solve).
        keep_factor=True adds "L" (n, (M+2)(M+3)/2), the packed factor with the right-hand-side rows, for mniw_trisolve.
        T0 (n, M, nvar) with P0 (M, nvar): an interface variable of nvar components (pgas_m_mniw_solve_n) -- "m" is then (n, nvar), "q"
        (n, nvar, nvar) and "L" (n, (M+1+nvar)(M+2+nvar)/2)."""
        n, M = T0.shape[0], T0.shape[1]
        nv = T0.shape[2] if T0.dim() == 3 else 0   # 0: the scalar layout (T0 (n, M))
        if nv:
            if P0.shape != (M, nv) or (R0 is not None and R0.shape != (M, nv)):
                raise ValueError("mniw_solve: P0 / R0 must be (M, nvar) when T0 is (n, M, nvar)")
            shp = {"m": (n, nv), "q": (n, nv, nv)}
            out = {k: torch.empty(shp.get(k, (n,)), dtype=torch.float64, device=self.device) for k in want}
        else:
            out = {k: self._vec(n) for k in want}
        if keep_factor:
            R = M + 1 + max(nv, 1)
            out["L"] = torch.empty((n, R * (R + 1) // 2), dtype=torch.float64, device=self.device)
            if nv:
                out["nvar"] = nv
        a = None if anc is None else anc.to(device=self.device, dtype=torch.int32).contiguous()
        for arr in (P0, P1, T0, T1, R0, R1, phi):
            if arr is not None and not (arr.is_contiguous() and arr.dtype == torch.float64 and arr.device == self.device):
                raise ValueError("mniw_solve: operands must be contiguous fp64 tensors on the engine's device")
        if a is not None and a.numel() != n:
            raise ValueError("mniw_solve: one ancestor index per particle expected")
        self.eng._chk(self.lib.pgas_m_mniw_solve_n(self.eng._h, n, M, max(nv, 1), float(scale), self._ptr(a), P0.data_ptr(), P1.data_ptr(), T0.data_ptr(),
                                                   T1.data_ptr(), self._ptr(R0), self._ptr(R1), self._ptr(phi), self._ptr(out.get("m")), self._ptr(out.get("c")),
                                                   self._ptr(out.get("q")), self._ptr(out.get("logdet")), self._ptr(out.get("L")), self.eng._stream()),
                      "pgas_m_mniw_solve")
        return out

    def mniw_trisolve(self, fac, anc, phi):
        """m = w[anc] . v, c = v . v with v = L[anc]^-1 phi for a factor kept by mniw_solve(keep_factor=True)."""
        n, M = phi.shape
        nv = int(fac.get("nvar", 0))
        R = M + 1 + max(nv, 1)
        a = None if anc is None else anc.to(device=self.device, dtype=torch.int32).contiguous()
        if fac["L"].shape != (n, R * (R + 1) // 2) or (a is not None and a.numel() != n):
            raise ValueError("mniw_trisolve: operand shapes do not match")
        m, c = (torch.empty((n, nv), dtype=torch.float64, device=self.device) if nv else self._vec(n)), self._vec(n)
        self.eng._chk(self.lib.pgas_m_mniw_trisolve_n(self.eng._h, n, M, max(nv, 1), self._ptr(a), fac["L"].data_ptr(), phi.contiguous().data_ptr(),
                                                      m.data_ptr(), c.data_ptr(), self.eng._stream()), "pgas_m_mniw_trisolve")
        return {"m": m, "c": c}

    def expr_eval(self, prog, state, inp, ivs, mode=0, anc=None, aux=None, mat=None, cR=0.0):
        """Run a traced model program (pgas_amd.exprs.Program) for every particle in one launch (pgas_m_expr_eval): mode 0 -> (n, n_out)
        values, 1 -> values + aux @ mat.T (draw_state), 2 -> cR - |mat (aux - values)|^2 / 2 per particle (log_likelihood)."""
        dev = self.device
        cache = prog.__dict__.setdefault("_dev", {})
        if dev not in cache:
            cache[dev] = (torch.as_tensor(prog.code.reshape(-1), dtype=torch.int32, device=dev), torch.as_tensor(prog.consts, dtype=torch.float64, device=dev))
        code_t, const_t = cache[dev]
        nx, nu, ivw = prog.widths
        n = state.shape[0]
        state = state.reshape(n, nx).contiguous()
        inp = inp.reshape(-1).contiguous() if nu else None
        ivs = [v.reshape(n, w).contiguous() for v, w in zip(ivs, ivw)]
        if state.dtype != torch.float64 or state.device != dev or len(ivs) != len(ivw) or (nu and inp.numel() != nu):
            raise ValueError("expr_eval: operands do not match the traced program")
        nout = len(prog.out_regs)
        out = torch.empty((n,) if mode == 2 else (n, nout), dtype=torch.float64, device=dev)
        a = None if anc is None else anc.to(device=dev, dtype=torch.int32).contiguous()
        regs = (C.c_int32 * nout)(*prog.out_regs)
        ptrs = (C.c_void_p * max(len(ivs), 1))(*[v.data_ptr() for v in ivs])
        wid = (C.c_int32 * max(len(ivs), 1))(*ivw)
        aux_c = None if aux is None else aux.reshape(-1).contiguous()
        mat_c = None if mat is None else mat.contiguous()
        self.eng._chk(self.lib.pgas_m_expr_eval(self.eng._h, n, code_t.data_ptr(), prog.code.shape[0], self._ptr(const_t) if prog.consts.size else None,
                                                int(prog.consts.size), prog.n_in, prog.n_reg, regs, nout, state.data_ptr(), nx, self._ptr(a),
                                                self._ptr(inp), nu, ptrs, wid, len(ivs), int(mode), self._ptr(aux_c), self._ptr(mat_c), float(cR),
                                                out.data_ptr(), self.eng._stream()), "pgas_m_expr_eval")
        return out

    def check(self):
        """Synchronises; raises if a matrix handed to mniw_solve since the last check was not positive definite."""
        self.eng._chk(self.lib.pgas_m_check(self.eng._h, self.eng._stream()), "pgas_m_check")

    def stats_gather_update(self, scale, anc, T, phi, xi):
        """T = (T0 (n,M), T1 (n,M,M), T2 (n,), T3 (n,)) -> scale * T[anc] + statistics of (xi, phi); new tensors."""
        T0, T1, T2, T3 = T
        n, M = T0.shape[0], T0.shape[1]
        nv = T0.shape[2] if T0.dim() == 3 else 1   # T0 (n, M, nvar), T2 (n, nvar, nvar), xi (n, nvar): several components
        out = (torch.empty_like(T0), torch.empty_like(T1), torch.empty_like(T2), torch.empty_like(T3))
        a = None if anc is None else anc.to(device=self.device, dtype=torch.int32).contiguous()
        if a is not None and a.numel() != n:
            raise ValueError("stats_gather_update: one ancestor index per particle expected")
        if phi.shape != (n, M) or xi.numel() != n * nv or T1.shape != (n, M, M) or T2.numel() != n * nv * nv:
            raise ValueError("stats_gather_update: operand shapes do not match (n, M[, nvar])")
        for arr in (T0, T1, T2, T3):
            if not arr.is_contiguous():
                raise ValueError("stats_gather_update: the statistics must be contiguous")
        self.eng._chk(self.lib.pgas_m_stats_gather_update_n(self.eng._h, n, M, nv, float(scale), self._ptr(a), T0.data_ptr(), T1.data_ptr(), T2.data_ptr(),
                                                            T3.data_ptr(), phi.contiguous().data_ptr(), xi.contiguous().data_ptr(), out[0].data_ptr(),
                                                            out[1].data_ptr(), out[2].data_ptr(), out[3].data_ptr(), self.eng._stream()),
                      "pgas_m_stats_gather_update")
        return out

    def weighted_stats(self, w, T):
        """sum_p w[p] T[p] for T = (T0 (n,M), T1 (n,M,M), T2 (n,), T3 (n,)) -> (S0 (M,), S1 (M,M), S2 (), S3 ())."""
        T0, T1, T2, T3 = T
        n, M = T0.shape[0], T0.shape[1]
        nv = T0.shape[2] if T0.dim() == 3 else 0
        w = w.contiguous()
        S0 = torch.empty((M, nv) if nv else (M,), dtype=torch.float64, device=self.device)
        S1 = torch.empty((M, M), dtype=torch.float64, device=self.device)
        k = max(nv, 1)
        S23 = torch.empty(k * k + 1, dtype=torch.float64, device=self.device)
        self.eng._chk(self.lib.pgas_m_weighted_stats_n(self.eng._h, n, M, k, w.data_ptr(), T0.data_ptr(), T1.data_ptr(), T2.data_ptr(), T3.data_ptr(),
                                                       S0.data_ptr(), S1.data_ptr(), S23.data_ptr(), S23.data_ptr() + 8 * k * k, self.eng._stream()),
                      "pgas_m_weighted_stats")
        return S0, S1, (S23[: k * k].reshape(k, k) if nv else S23[0]), S23[k * k]

    def systematic_resample(self, u, logw):
        return self.eng.systematic_resample(u, logw)
