"""not-gpu: host/htk_lattice.h -- the FST -> HTK Standard Lattice Format conversion CRFDecode runs for htk_lat_outdir
(CRFDecode/src/Main.cpp:432-720) -- on hand-worked machines: a best-path chain, a frame-synchronous lattice with two
competing words, and the three inputs the reference stops on (epsilon input label, a state inside two different
words, a state reached after different numbers of arcs).  The expected files below were worked out by hand from the
reference's description of the walk (word arcs run from the state where a word's first arc leaves to the state where
the next word's first arc leaves, or to a state without arcs; weights are the negated sums; time = arcs from the
start state, written 0.01 (time + 1) with the start state at -1)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    lib = os.path.join(ROOT, "asr-craft_amd", "lib")
    if not os.path.exists(os.path.join(lib, "libcrf_amd_host.so")):
        pytest.fail("libcrf_amd_host.so not built: run __graft_entry__.build()")
    out = str(tmp_path_factory.mktemp("htk") / "htk_lattice_check")
    r = subprocess.run(["g++", "-O1", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "asr-craft_amd", "host"),
                        os.path.join(ROOT, "tests", "host", "htk_lattice_check.cpp"), "-o", out, "-L" + lib, "-Wl,-rpath," + lib,
                        "-lcrf_amd_host", "-lscrf_amd"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    return out


HEADER = "VERSION=1.0\nUTTERANCE=%s\nlmscale=1.00  wdpenalty=0.00\nprscale=1.00\nacscale=1.00\n"


def convert(exe, tmp_path, fst_text, name="utt1", symbols="<eps> 0\nA 11\nB 12\nC 13\n"):
    f, sy, out = str(tmp_path / "m.txt"), str(tmp_path / "sym.txt"), str(tmp_path / "out.slf")
    open(f, "w").write(fst_text)
    if symbols is not None:
        open(sy, "w").write(symbols)
    r = subprocess.run([exe, f, sy if symbols is not None else "-", name, out], capture_output=True, text=True, timeout=60)
    return r, (open(out).read() if os.path.exists(out) else None)


def test_best_path_chain_becomes_one_arc_per_word(exe, tmp_path):
    # word A over three arcs (phones 1 1 2), word B over two (3 3); final weight is not part of any arc
    fst = "0 1 1 11 0.5\n1 2 1 0 0.25\n2 3 2 0 1.5\n3 4 3 12 2\n4 5 3 0 0.125\n5 7.5\n"
    r, slf = convert(exe, tmp_path, fst)
    assert r.returncode == 0 and r.stdout.strip() == "nodes 3 arcs 2", r.stdout + r.stderr
    # nodes in the order they were needed: state 0 (start of A), state 3 (start of B: A's arcs end there), state 5 (no arcs)
    assert slf == HEADER % "utt1" + ("N=3 L=2\n"
                                     "I=0 t=0 W=!NULL\n"
                                     "I=1 t=0.03 W=A v=1\n"
                                     "I=2 t=0.05 W=B v=1\n"
                                     "J=0 S=0 E=1 a=-2.25 l=-0 r=0.00\n"
                                     "J=1 S=1 E=2 a=-2.125 l=-0 r=0.00\n")


def test_frame_lattice_with_two_competing_words(exe, tmp_path):
    # from the start state word A (2 arcs: 0 -> 1 -> 3) or word B (2 arcs: 0 -> 2 -> 3); then word C (3 -> 4); equal lengths
    fst = ("0 1 1 11 1\n0 2 2 12 2\n"
           "1 3 1 0 0.5\n"
           "2 3 2 0 0.25\n"
           "3 4 3 13 4\n"
           "4 0\n")
    r, slf = convert(exe, tmp_path, fst)
    # state 3 lies in word A by its first visitor (state 1): the arc from state 2 arrives inside word B -> the reference's error
    assert r.returncode == 3 and "two incoming arcs going through the fst state 3 with different word labels: 11 and 12" in r.stderr
    # the same shape with ONE word identity on both branches (two pronunciations of A) converts: two parallel word arcs
    fst = fst.replace("0 2 2 12 2", "0 2 2 11 2")
    r, slf = convert(exe, tmp_path, fst, name="two_prons")
    assert r.returncode == 0 and r.stdout.strip() == "nodes 3 arcs 3", r.stdout + r.stderr
    assert slf == HEADER % "two_prons" + ("N=3 L=3\n"
                                          "I=0 t=0 W=!NULL\n"
                                          "I=1 t=0.02 W=A v=1\n"
                                          "I=2 t=0.03 W=C v=1\n"
                                          "J=0 S=0 E=1 a=-1.5 l=-0 r=0.00\n"
                                          "J=1 S=0 E=1 a=-2.25 l=-0 r=0.00\n"
                                          "J=2 S=1 E=2 a=-4 l=-0 r=0.00\n")


def test_inputs_the_reference_stops_on(exe, tmp_path):
    r, _ = convert(exe, tmp_path, "0 1 1 11 1\n1 2 0 12 1\n2 0\n")
    assert r.returncode == 3 and "currently doesn't support epsilon input labels on any fst arc" in r.stderr
    # state 2 is reached after one arc (0 -> 2) and after two (0 -> 1 -> 2): a segmental lattice with durations 1 and 2
    r, _ = convert(exe, tmp_path, "0 1 1 11 1\n0 2 1 11 1\n1 2 1 0 1\n2 0\n")
    assert r.returncode == 3 and "two paths reach the fst state 2 at different time frame: 0 and 1" in r.stderr
    # a node inside a word without an output symbol table
    r, _ = convert(exe, tmp_path, "0 1 1 11 1\n1 0\n", symbols=None)
    assert r.returncode == 3 and "output symbol table has not been set" in r.stderr
    # an unknown word id prints as an empty label (SymbolTable::Find of a missing key)
    r, slf = convert(exe, tmp_path, "0 1 1 99 1\n1 0\n")
    assert r.returncode == 0 and "I=1 t=0.01 W= v=1\n" in slf
