"""Declarative stand-ins for the Python callables the reference hands to its algorithms.

The reference's ``basis_fcn(state, input)`` and ``likelihood_fcn(obs, state, input)`` are Python
lambdas traced by JAX (src/PGAS.py:20-21).  A HIP kernel cannot trace a lambda, so the two families
the reference instantiates are described by small tables (see include/pgas_hip.h).  The objects
are still callable on the host, so code such as ``len(basis_fcn(m0, u0))`` (src/PGAS.py:41-43)
keeps working.
"""
from __future__ import annotations

import numpy as np


class HilbertBasis:
    """Hilbert-space GP basis on a box (reference src/BasisFunctions.py:8-80).

    phi_m(x) = prod_d sqrt(1/L_d) sin(sqrt(eig_md) (x_d - center_d + L_d)),  sqrt(eig_md) = pi j_md / (2 L_d).
    """

    def __init__(self, indices, domain_boundary):
        box = np.atleast_2d(np.asarray(domain_boundary, dtype=np.float64))
        self.indices = np.ascontiguousarray(np.atleast_2d(indices), dtype=np.int32)  # (M, D) frequencies j
        self.size = box[:, 1] - box[:, 0]
        self.center = (box[:, 0] + box[:, 1]) / 2
        self.L = self.size / 2
        self.M, self.D = self.indices.shape
        if self.D != box.shape[0]:
            raise ValueError("index table and domain have different dimensions")

    @property
    def eigenvalues(self):
        return (np.pi * self.indices / self.size) ** 2

    @property
    def norm(self):
        return float(np.prod(np.sqrt(1.0 / self.L)))

    def __call__(self, x):
        """Host evaluation for one point x (D,) or scalar when D == 1 -> (M,)."""
        xc = np.asarray(x, dtype=np.float64).reshape(-1)[: self.D] - self.center
        ang = np.pi * self.indices * (xc + self.L) / self.size
        return np.prod(np.sqrt(1.0 / self.L) * np.sin(ang), axis=1)

    def on(self, sel, div=None):
        """basis evaluated at concat(state, input)[sel] / div -- a ``basis_fcn(state, input)``."""
        return BasisMap(self, sel, div)

    def __len__(self):
        return self.M


class BasisMap:
    """``basis_fcn(state, input) = basis(concat(state, input)[sel] / div)``.

    Covers src/Toy_Example.py:146 (sel=[0]), src/SingleMassOscillator.py:151 (sel=[0,1]) and
    src/EMPS.py:110-113 (sel=[0,1,2], div=[0.4,0.4,160]).
    """

    def __init__(self, basis: HilbertBasis, sel, div=None):
        self.basis = basis
        self.sel = np.ascontiguousarray(np.atleast_1d(sel), dtype=np.int32)
        self.div = np.ones(basis.D) if div is None else np.asarray(div, dtype=np.float64).reshape(-1)
        if self.sel.shape[0] != basis.D or self.div.shape[0] != basis.D:
            raise ValueError("sel/div must have one entry per basis dimension")

    def __call__(self, state, input=None):
        v = np.atleast_1d(np.asarray(state, dtype=np.float64)).reshape(-1)
        if input is not None and np.size(input):
            v = np.concatenate([v, np.atleast_1d(np.asarray(input, dtype=np.float64)).reshape(-1)])
        return self.basis(v[self.sel] / self.div)

    def bind(self, ops):
        """Algorithm1 / Algorithm3 hand over their device operations: batch() then runs as one HIP launch."""
        self._ops = ops

    def trajectory(self, states, inputs=None):
        """Basis along a trajectory in one call: states (T, n_x), inputs (T, n_u) -- row t is evaluated with inputs[t] -> (T, M)."""
        return self.batch(states, inputs, per_row=True)

    def batch(self, state, input=None, per_row=False):
        """Batched evaluation for the marginalised family (Algorithm1/2/3): state (N, n_x) NumPy array or torch tensor, input (n_u,)
        (one input for all rows; per_row: (N, n_u), one per row) or None -> (N, M) of the same kind.  phi = prod_d sqrt(1/L_d) sin(pi j_d (v_d/div_d - center_d + L_d) / size_d)
        (src/BasisFunctions.py:77-80)."""
        b = self.basis
        if isinstance(state, np.ndarray):
            v = state.reshape(state.shape[0], -1)
            if input is not None and np.size(input):
                inp = np.asarray(input, dtype=np.float64)
                rows = inp.reshape(v.shape[0], -1) if inp.ndim == 2 and inp.shape[0] == v.shape[0] and per_row else np.broadcast_to(inp.reshape(1, -1), (v.shape[0], inp.size))
                v = np.concatenate([v, rows], axis=1)
            ang = np.pi * b.indices[None, :, :] * ((v[:, self.sel] / self.div - b.center + b.L) / b.size)[:, None, :]
            return np.prod(np.sqrt(1.0 / b.L) * np.sin(ang), axis=2)
        import torch

        ops = getattr(self, "_ops", None)
        if ops is not None and not per_row and b.D <= 4 and state.dtype == torch.float64 and state.device == ops.device:
            return ops.hilbert_basis(self, state, input)   # one HIP launch (pgas_m_hilbert_basis) instead of ten torch launches
        v = state.reshape(state.shape[0], -1)
        if input is not None and input.numel():
            rows = input.reshape(v.shape[0], -1) if per_row else input.reshape(1, -1).expand(v.shape[0], -1)
            v = torch.cat([v, rows], dim=1)
        key = v.device
        if getattr(self, "_tcache", None) is None or self._tcache[0] != key:
            tt = lambda a: torch.as_tensor(np.asarray(a, dtype=np.float64), device=v.device)  # noqa: E731
            self._tcache = (key, torch.as_tensor(self.sel.astype(np.int64), device=v.device), tt(self.div), tt(b.center), tt(b.L), tt(b.size),
                            tt(b.indices.astype(np.float64)), tt(np.sqrt(1.0 / b.L)))
        _, sel, div, center, L, size, idx, amp = self._tcache
        ang = np.pi * idx[None, :, :] * ((v[:, sel] / div - center + L) / size)[:, None, :]
        return torch.prod(amp * torch.sin(ang), dim=2)

    # engine tables: r_d = v[sel_d] * alpha_d + beta_d
    @property
    def alpha(self):
        return 1.0 / (self.div * self.basis.size)

    @property
    def beta(self):
        return (self.basis.L - self.basis.center) / self.basis.size


class GaussianLikelihood:
    """``likelihood_fcn(obs, state, input) = log N(obs; H state, R)``
    (src/Toy_Example.py:142-144, src/EMPS.py:250-252: mvn.logpdf(obs, mean=f_y(state), cov=R))."""

    def __init__(self, H, R):
        self.H = np.atleast_2d(np.asarray(H, dtype=np.float64))
        self.R = np.atleast_2d(np.asarray(R, dtype=np.float64))
        self.ny, self.nx = self.H.shape
        if self.R.shape != (self.ny, self.ny):
            raise ValueError("R must be (ny, ny)")
        self.LR = np.linalg.cholesky(self.R)
        self.LRinv = np.linalg.inv(self.LR)
        self.cR = float(-0.5 * self.ny * np.log(2 * np.pi) - np.sum(np.log(np.diag(self.LR))))

    @classmethod
    def of_component(cls, index, nx, R):
        """f_y(x) = x[index] (src/EMPS.py:197-198, src/SingleMassOscillator.py:47-48)."""
        H = np.zeros((1, nx))
        H[0, index] = 1.0
        return cls(H, R)

    def __call__(self, obs, state, input=None):
        e = np.atleast_1d(np.asarray(obs, dtype=np.float64)) - self.H @ np.atleast_1d(np.asarray(state, dtype=np.float64))
        w = self.LRinv @ e
        return float(self.cR - 0.5 * w @ w)
