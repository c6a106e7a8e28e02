"""-m "not gpu": the L-BFGS optimiser behind CRF_LBFGSTrainer (asr-craft_amd/host/lbfgs.h), PINNED to the reference.

The reference links its vendored libLBFGS (CRF/src/utils/lbfgs.c) with default parameters
(trainers/CRF_LBFGSTrainer.cpp:80).  That file is plain C and compiles here from where it lies
(oracle/Makefile -> oracle/_ref/liblbfgs_ref.so); tests/golden/gen_lbfgs_golden.py ran it over the four problems
of tests/host/lbfgs_problems.h and committed every accepted iterate as tests/golden/lbfgs_ref.npz.
  * fixture test: the iterates of lbfgs.h (x_k, f_k, step, line-search count, norms, per iteration; return code and
    evaluation count) against those vectors, to 1e-12 -- they are in fact bit-identical on this toolchain;
  * live test (only where oracle/_ref exists, i.e. in the build container): the two programs' outputs byte for byte;
  * the optimiser's own contract: closed-form minima, the stopping rule, never uphill, the stop callback."""
import importlib.util
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "asr-craft_amd", "host")
spec = importlib.util.spec_from_file_location("gen_lbfgs_golden", os.path.join(ROOT, "tests", "golden", "gen_lbfgs_golden.py"))
gen = importlib.util.module_from_spec(spec)
spec.loader.exec_module(gen)


def build(tmp, src, name):
    exe = str(tmp / name)
    subprocess.run(["g++", "-O2", "-ffp-contract=off", "-std=c++17", "-Wall", "-Werror", "-I" + HOST,
                    "-I" + os.path.join(ROOT, "tests", "host"), os.path.join(ROOT, "tests", "host", src), "-o", exe],
                   check=True, timeout=300)
    return exe


@pytest.fixture(scope="module")
def trace_text(tmp_path_factory):
    exe = build(tmp_path_factory.mktemp("lbfgs_trace"), "lbfgs_trace.cpp", "lbfgs_trace")
    return subprocess.run([exe], capture_output=True, text=True, check=True, timeout=120).stdout


def test_iterates_equal_the_reference_librarys_golden_vectors(trace_text):
    mine, codes = gen.parse_trace(trace_text)
    gold = np.load(os.path.join(ROOT, "tests", "golden", "lbfgs_ref.npz"))
    # the status codes are the library's numeric values (utils/lbfgs.h:75-145)
    assert list(codes) == list(gold["codes"])
    assert sorted(mine) == ["lse50", "quadratic", "rosenbrock", "steep10"]
    for name, d in mine.items():
        assert [d["ret"], d["evals"]] == list(gold[name + "_end"]), name
        assert list(d["k"]) == list(gold[name + "_k"]), name
        assert list(d["ls"]) == list(gold[name + "_ls"]), name       # evaluations per line search
        for key in ("step", "fx", "xnorm", "gnorm"):
            np.testing.assert_allclose(np.array(d[key]), gold[name + "_" + key], rtol=1e-12, atol=0, err_msg=name + " " + key)
        x = np.array(d["x"])
        np.testing.assert_allclose(x, gold[name + "_x"], rtol=0, atol=1e-12 * max(1.0, np.abs(gold[name + "_x"]).max()), err_msg=name)
        assert d["fx_end"] == pytest.approx(float(gold[name + "_fx_end"][0]), rel=1e-12)
    # the problems do exercise the search: several evaluations per search occur, and hundreds of iterations
    assert max(mine["rosenbrock"]["ls"]) >= 5 and len(mine["quadratic"]["k"]) > 200


@pytest.mark.skipif(not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "liblbfgs_ref.so")) or not os.path.isdir(gen.REF),
                    reason="the reference's libLBFGS is only built where /root/reference exists")
def test_iterates_equal_the_reference_library_run_live(trace_text):
    assert gen.run_reference_trace() == trace_text      # every iterate, hex floats, byte for byte


@pytest.fixture(scope="module")
def lines(tmp_path_factory):
    exe = build(tmp_path_factory.mktemp("lbfgs"), "lbfgs_check.cpp", "lbfgs_check")
    out = subprocess.run([exe], capture_output=True, text=True, check=True, timeout=60).stdout
    res = {}
    for ln in out.splitlines():
        f = ln.split()
        if "ret=" not in ln:
            res[f[0]] = int(f[1])
            continue
        d = {k: float(v) for k, v in (t.split("=") for t in f[1:5])}
        d["x"] = np.array([float(v) for v in f[6:]])
        res[f[0]] = d
    return res


def test_rosenbrock_reaches_the_minimum_and_never_goes_uphill(lines):
    r = lines["rosenbrock"]
    assert r["ret"] == 0 and r["fx"] < 1e-10
    np.testing.assert_allclose(r["x"], 1.0, atol=1e-5)
    assert lines["rosenbrock_monotone"] == 1


def test_ill_conditioned_quadratic_converges_to_its_closed_form(lines):
    r = lines["quadratic"]
    assert r["ret"] == 0
    m = np.arange(12) - 5.5
    # stopping rule |g| / max(1,|x|) <= 1e-5 (the library's default epsilon): |c (x - m)| is below 1e-5 |x|
    c = 10.0 ** (np.arange(12) / 3.0)
    assert np.linalg.norm(c * (r["x"] - m)) <= 1e-5 * max(1.0, np.linalg.norm(r["x"])) * (1 + 1e-9)
    np.testing.assert_allclose(r["x"], m, atol=1e-4)


def test_start_at_the_minimum_and_progress_stop(lines):
    assert lines["at_minimum"]["ret"] == 2 and lines["at_minimum"]["evals"] == 1
    s = lines["stopped"]
    assert s["ret"] == 1 and s["iters"] == 3
    x = s["x"]
    k = 0.3 * (np.arange(4) + 1)
    assert s["fx"] == pytest.approx(float((np.exp(k * x) - x).sum()), rel=1e-14)    # fx belongs to the returned x
    assert s["fx"] < float((np.exp(k * 3.0) - 3.0).sum())
