#!/usr/bin/env python3
"""Generates tests/golden/lbfgs_ref.npz: every accepted iterate of the REFERENCE's vendored libLBFGS
(/root/reference/CRF/src/utils/lbfgs.c compiled where it lies into oracle/_ref/liblbfgs_ref.so by oracle/Makefile,
default parameters as trainers/CRF_LBFGSTrainer.cpp:80 passes them) on the four problems of
tests/host/lbfgs_problems.h.  Runs in the build container only (the reference tree does not travel); the fixture and
this script are committed.   usage: python tests/golden/gen_lbfgs_golden.py"""
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("SCRF_REFERENCE", "/root/reference")


def parse_trace(text):
    """{problem: dict(k, ls, step, fx, xnorm, gnorm, x[iters][n], ret, evals, fx_end)}, codes"""
    out, codes = {}, None
    for ln in text.splitlines():
        f = ln.split()
        if f[0] == "codes":
            codes = np.array([int(v) for v in f[1:]])
        elif f[0] == "it":
            d = out.setdefault(f[1], {"k": [], "ls": [], "step": [], "fx": [], "xnorm": [], "gnorm": [], "x": []})
            d["k"].append(int(f[2])); d["ls"].append(int(f[3]))
            for key, v in zip(("step", "fx", "xnorm", "gnorm"), f[4:8]):
                d[key].append(float.fromhex(v))
            d["x"].append([float.fromhex(v) for v in f[8:]])
        elif f[0] == "end":
            d = out[f[1]]
            d["ret"], d["evals"], d["fx_end"] = int(f[2]), int(f[3]), float.fromhex(f[4])
    return out, codes


def run_reference_trace():
    lib = os.path.join(ROOT, "oracle", "_ref")
    if not os.path.exists(os.path.join(lib, "liblbfgs_ref.so")):
        raise SystemExit("oracle/_ref/liblbfgs_ref.so is missing: run `make -C oracle` where /root/reference exists")
    with tempfile.TemporaryDirectory() as td:
        exe = os.path.join(td, "lbfgs_ref_trace")
        subprocess.run(["gcc", "-O2", "-ffp-contract=off", "-I" + os.path.join(REF, "CRF", "src", "utils"),
                        "-I" + os.path.join(ROOT, "tests", "host"), os.path.join(HERE, "lbfgs_ref_trace.c"), "-o", exe,
                        "-L" + lib, "-llbfgs_ref", "-Wl,-rpath," + lib, "-lm"], check=True)
        return subprocess.run([exe], capture_output=True, text=True, check=True).stdout


def main():
    tr, codes = parse_trace(run_reference_trace())
    arrays = {"codes": codes}
    for name, d in tr.items():
        for key in ("k", "ls"):
            arrays["%s_%s" % (name, key)] = np.array(d[key], dtype=np.int32)
        for key in ("step", "fx", "xnorm", "gnorm", "x"):
            arrays["%s_%s" % (name, key)] = np.array(d[key], dtype=np.float64)
        arrays["%s_end" % name] = np.array([d["ret"], d["evals"]], dtype=np.int64)
        arrays["%s_fx_end" % name] = np.array([d["fx_end"]])
    np.savez_compressed(os.path.join(HERE, "lbfgs_ref.npz"), **arrays)
    print("wrote lbfgs_ref.npz:", {k: len(v["k"]) for k, v in tr.items()})


if __name__ == "__main__":
    sys.exit(main())
