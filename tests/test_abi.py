"""The C-ABI library builds, loads (no GPU needed) and exports every symbol include/pgas_hip.h declares."""
import ctypes
import os
import re

from common import ROOT


def _declared():
    names = set()
    for h in ("pgas_hip.h", "pgas_marginal.h"):   # the two headers that declare entry points (the others are inline arithmetic)
        txt = open(os.path.join(ROOT, "include", h)).read()
        txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
        names |= set(re.findall(r"\b(pgas_[a-z_0-9]+)\s*\(", txt))
    return sorted(names)


def test_header_symbols_are_exported():
    from pgas_amd import _lib

    assert os.path.exists(_lib.LIB_PATH), "run `python __graft_entry__.py` first"
    L = ctypes.CDLL(_lib.LIB_PATH)
    names = _declared()
    assert len(names) >= 13
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/pgas_hip.h but not exported"
    assert set(_lib.EXPORTS) == set(names), "pgas_amd/_lib.py EXPORTS out of sync with the header"


def test_segment_size_matches_oracle():
    from oracle import canon
    from pgas_amd import _lib

    assert _lib.load().pgas_segment_size() == canon.lib().oc_seg() == 1024


def test_engine_fails_loudly_without_gpu():
    import pytest
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import pgas_amd
    from pgas_amd import experiments
    from pgas_amd._lib import PgasError

    pb = experiments.smo_pgas(T=4)
    with pytest.raises(PgasError, match="no CPU path"):
        pgas_amd.condSequentialMonteCarlo(16, pb.observations, pb.inputs, pb.init_state_mean, pb.init_state_cov, pb.likelihood_fcn, pb.basis_fcn)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "bayesian-inference-with-explicit-and-implicit-prior-knowledge_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("oracle/pgas_canon.c", "").replace("canonical C oracle", "").replace("the oracle", "").lower() or f == "pgas_kernels.hip.h", f
