"""eiquadprog's own two-variable unit problems for EiquadprogFast (tests/golden/eiquadprog/make_eiquadprog_unit.py: optimum, objective and outcome of each
derived by hand there) against the CPU restatement `wbco_eiquadprog_fast` and, on the GPU, against the dense seam `wbcqp_solve_dense_host`:
one more pin of the status map (eiquadprog code -> tsid code) that does not pass through the oracle's own outputs."""
import os

import numpy as np
import pytest

Z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "eiquadprog", "eiquadprog_unit.npz"))
NAMES = [str(n) for n in Z["names"]]


def _qp(nm):
    return [Z[nm + k] for k in ("_H", "_g", "_CE", "_ce0", "_CI", "_ci0")]


@pytest.mark.parametrize("nm", NAMES)
def test_oracle_on_eiquadprogs_unit_problems(oracle_mod, nm):
    r = oracle_mod.eiquadprog(*_qp(nm))
    assert r["status"] == int(Z[nm + "_status_eiquadprog"]), (nm, r["status"])
    if r["status"] == 0:
        assert np.abs(r["x"] - Z[nm + "_x"]).max() <= 1e-12
        assert abs(r["fval"] - float(Z[nm + "_f"])) <= 1e-12


def test_fixture_is_what_its_generator_writes(tmp_path):
    """the committed .npz equals a fresh run of the committed script"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("mk", os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "eiquadprog", "make_eiquadprog_unit.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    assert [c[0] for c in mk.CASES] == NAMES
    for name, Q, C, Aeq, Beq, Ain, Bin, xs, fs, st_e, st_t in mk.CASES:
        assert np.array_equal(Z[name + "_H"], np.asarray(Q, float)) and np.array_equal(Z[name + "_CI"], np.asarray(Ain, float).reshape(-1, 2))
        assert int(Z[name + "_status_tsid"]) == st_t
        if xs is not None:
            # the hand-derived optimum is a KKT point: feasibility, and the objective value written next to it
            x = np.asarray(xs, float)
            assert np.abs(np.asarray(Aeq, float).reshape(-1, 2) @ x + np.asarray(Beq, float)).max(initial=0.0) <= 1e-15
            assert (np.asarray(Ain, float).reshape(-1, 2) @ x + np.asarray(Bin, float)).min(initial=0.0) >= -1e-15
            assert abs(0.5 * x @ np.asarray(Q, float) @ x + np.asarray(C, float) @ x - fs) <= 1e-15


@pytest.mark.gpu
@pytest.mark.parametrize("nm", NAMES)
def test_dense_seam_on_eiquadprogs_unit_problems(nm):
    from inria_wbc_amd import capi
    H, g, CE, ce0, CI, ci0 = _qp(nm)
    h = capi.Handle(0, capi.F64)
    got = h.solve_dense_host(H, g, CE if CE.size else None, ce0 if CE.size else None, CI if CI.size else None, ci0 if CI.size else None)
    h.close()
    assert got["status"][0] == int(Z[nm + "_status_tsid"]), (nm, got["status"][0])
    if got["status"][0] == 0:
        assert np.abs(got["x"][0] - Z[nm + "_x"]).max() <= 1e-12
        assert abs(got["objective"][0] - float(Z[nm + "_f"])) <= 1e-12
