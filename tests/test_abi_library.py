"""not-gpu: the C-ABI library builds, loads and exports every symbol include/scrf_abi.h declares;
without a GPU the engine refuses to run (no CPU fallback in the product path)."""
import ctypes as C
import os
import re

import pytest

import scrf_amd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "scrf_abi.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(scrf_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported():
    if not os.path.exists(scrf_amd.lib_path()):
        import __graft_entry__ as g
        g.build()
    lib = scrf_amd.load_library()
    syms = declared_symbols()
    assert len(syms) >= 30
    for s in syms:
        assert hasattr(lib, s), s


def test_no_cpu_fallback_without_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(scrf_amd.ScrfError) as ei:
        scrf_amd.Engine(scrf_amd.make_config(L=3, D=2, F=18))
    assert ei.value.code == 2  # SCRF_ERR_NO_DEVICE


def test_product_never_references_the_oracle():
    """the product tree must not import, link or execute anything under oracle/"""
    bad = []
    for base, _, files in os.walk(os.path.join(ROOT, "asr-craft_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", "Makefile")):
                txt = open(os.path.join(base, f), errors="ignore").read()
                if re.search(r"oracle|orc_|libscrf_oracle", txt):
                    bad.append(os.path.join(base, f))
    assert not bad, bad


def test_host_classes_keep_the_reference_shaped_signatures(tmp_path):
    """tests/host/interface_conformance.cpp carries static_asserts on constructor argument lists and member-function
    pointer types of asr-craft_amd/host/crf_amd.h (the reference interfaces of SURVEY 8b); compiling it is the check.
    (Linking and running it needs the GPU: tests/test_gpu_cli.py.)"""
    import subprocess
    r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
                        "-I" + os.path.join(ROOT, "asr-craft_amd", "host"),
                        os.path.join(ROOT, "tests", "host", "interface_conformance.cpp")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
