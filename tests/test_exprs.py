"""pgas_amd.exprs (CPU): models written against an array namespace trace into register programs that reproduce the NumPy callables."""
import numpy as np
import pytest

from common import experiments
from pgas_amd import exprs


def _cases():
    return {"smo": experiments.smo_marginal, "vehicle": experiments.vehicle_marginal, "emps": experiments.emps_marginal, "toy": experiments.toy_marginal,
            "smo2": experiments.smo_two_component_marginal}


@pytest.mark.parametrize("name", ["smo", "vehicle", "emps", "toy", "smo2"])
def test_traced_programs_reproduce_the_numpy_models(name):
    pb = _cases()[name](T=5)
    nx = pb.init_state_mean.shape[0]
    U = np.asarray(pb.inputs, dtype=np.float64).reshape(pb.T, -1)
    ivw = [np.asarray(m).reshape(-1).shape[0] for m in pb.init_int_var_mean]
    rng = np.random.default_rng(1)
    N = 64
    st, u, ivs = rng.standard_normal((N, nx)) * 0.4, U[2] + 0.1, [rng.standard_normal((N, w)) * 0.5 for w in ivw]
    fn = pb.model(np)
    for which in (0, 1):
        def traced(state, inp, *iv, _w=which):
            return pb.model(exprs.SymNamespace(state.tr))[_w](state, inp, *iv)
        prog = exprs.trace(traced, nx, U.shape[1], ivw)
        assert prog.n_reg <= exprs.MAX_REG and prog.code.shape[1] == 4 and prog.code[:, 0].min() >= 1 and prog.code[:, 0].max() <= 14
        ref = np.asarray(fn[which](st, u, *ivs)).reshape(N, -1)
        assert np.array_equal(exprs.run_numpy(prog, st, u, ivs), ref)   # same operations, common subexpressions shared: same bits
    # the RK4 transition of the oscillator collapses to a handful of instructions (its force is constant over the step)
    if name == "smo":
        prog = exprs.trace(lambda s, i, *v: pb.model(exprs.SymNamespace(s.tr))[0](s, i, *v), nx, U.shape[1], ivw)
        assert prog.code.shape[0] < 30


def test_unsupported_constructs_are_refused_at_trace_time():
    with pytest.raises(TypeError):
        exprs.trace(lambda s, u: s @ s, 2, 1, [])                       # matrix product
    with pytest.raises(TypeError):
        exprs.trace(lambda s, u: s[:, 0] + s, 2, 1, [])                 # (N,) against (N, k)
    with pytest.raises(TypeError):
        exprs.trace(lambda s, u: np.cos(u), 2, 1, [])                   # a uniform result, not per particle
