"""-m "not gpu": the L-BFGS optimiser behind CRF_LBFGSTrainer (asr-craft_amd/host/lbfgs.h).  The reference links
libLBFGS with its default parameters (trainers/CRF_LBFGSTrainer.cpp:55-62); libLBFGS is not in this image, so the
checks are closed-form minima, the stopping rule, and scipy's L-BFGS on the same functions."""
import os
import subprocess

import numpy as np
import pytest
from scipy.optimize import minimize

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lines(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("lbfgs") / "lbfgs_check")
    subprocess.run(["g++", "-O2", "-std=c++17", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "asr-craft_amd", "host"),
                    os.path.join(ROOT, "tests", "host", "lbfgs_check.cpp"), "-o", exe], check=True, timeout=300)
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

    def f(p):
        t1 = 1 - p[0::2]; t2 = 10 * (p[1::2] - p[0::2] ** 2)
        g = np.zeros_like(p); g[1::2] = 20 * t2; g[0::2] = -2 * (p[0::2] * g[1::2] + t1)
        return float((t1 ** 2 + t2 ** 2).sum()), g
    x0 = np.tile([-1.2, 1.0], 10)
    s = minimize(f, x0, jac=True, method="L-BFGS-B", options={"maxcor": 6, "gtol": 1e-8, "ftol": 1e-15})
    np.testing.assert_allclose(r["x"], s.x, atol=1e-4)
    assert r["evals"] <= 2 * s.nfev + 10      # the same order of work as scipy's implementation


def test_ill_conditioned_quadratic_converges_to_its_closed_form(lines):
    r = lines["quadratic"]
    assert r["ret"] == 0
    m = np.arange(12) - 5.5
    # stopping rule |g| / max(1,|x|) <= 1e-5 (libLBFGS default epsilon): |c (x - m)| is below 1e-5 |x|
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
