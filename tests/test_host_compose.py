"""not-gpu: crf_amd::composeShortestPath (libcrf_amd_host.so) -- the Compose + ShortestPath + RmEpsilon +
TopSort of CRFFstDecode's MLF path (CRFFstDecode/src/Main.cpp:1002-1023) in one host pass -- against an
exhaustive enumeration of every (lattice path, LM path) pair on random small machines: acyclic lattices with
epsilon-output arcs, LMs with epsilon-input arcs (carrying output symbols), parallel arcs, missing labels,
several final states.  OpenFST itself is not in the tree: the tie order is unpinned, so totals are compared
to float tolerance and the label sequences wherever the optimum is unique."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    lib = os.path.join(ROOT, "asr-craft_amd", "lib")
    if not os.path.exists(os.path.join(lib, "libcrf_amd_host.so")):
        pytest.fail("libcrf_amd_host.so not built: run __graft_entry__.build()")
    out = str(tmp_path_factory.mktemp("compose") / "compose_best_path")
    r = subprocess.run(["g++", "-O1", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "asr-craft_amd", "host"),
                        os.path.join(ROOT, "tests", "host", "compose_best_path.cpp"), "-o", out, "-L" + lib, "-Wl,-rpath," + lib,
                        "-lcrf_amd_host", "-lscrf_amd"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    return out


def write_fst(path, arcs, finals, start):
    with open(path, "w") as f:
        arcs = sorted(arcs, key=lambda a: a[0] != start)      # text format: the first line's source is the start state
        for a in arcs:
            f.write("%d %d %d %d %.9g\n" % a)
        for s, w in finals.items():
            f.write("%d %.9g\n" % (s, w))


def brute(lat, lfin, lstart, lm, mfin, mstart, max_eps=3):
    """every lattice path x every LM walk that reads its output labels (at most max_eps epsilon-input LM arcs in a
    row); returns sorted [(total, ilabels, olabels)]"""
    lout = {}
    for a in lat:
        lout.setdefault(a[0], []).append(a)
    mout = {}
    for a in lm:
        mout.setdefault(a[0], []).append(a)
    res = []

    def lm_walks(q, labels, k, cost, outs, eps_run):
        if k == len(labels):
            if q in mfin:
                yield cost + mfin[q], outs
        for a in mout.get(q, []):
            if a[2] == 0 and eps_run < max_eps:
                yield from lm_walks(a[1], labels, k, cost + a[4], outs + ([a[3]] if a[3] else []), eps_run + 1)
            elif k < len(labels) and a[2] == labels[k]:
                yield from lm_walks(a[1], labels, k + 1, cost + a[4], outs + ([a[3]] if a[3] else []), 0)

    def lat_paths(s, cost, ils, ols):
        if s in lfin:
            for c, outs in lm_walks(mstart, [o for o in ols if o], 0, 0.0, [], 0):
                res.append((cost + lfin[s] + c, [i for i in ils if i], outs))
        for a in lout.get(s, []):
            lat_paths(a[1], cost + a[4], ils + [a[2]], ols + [a[3]])
    lat_paths(lstart, 0.0, [], [])
    return sorted(res, key=lambda r: r[0])


@pytest.mark.parametrize("seed", range(12))
def test_compose_shortest_path_equals_exhaustive_enumeration(exe, tmp_path, seed):
    rng = np.random.RandomState(900 + seed)
    # acyclic lattice over states 0..S-1 (ids are a topological order), labels 1..3, some epsilon-output arcs
    S = int(rng.randint(4, 8))
    lat = []
    for s in range(S - 1):
        for _ in range(int(rng.randint(1, 4))):
            d = int(rng.randint(s + 1, min(S, s + 3)))
            il = int(rng.randint(1, 4))
            ol = il if rng.rand() > 0.25 else 0
            lat.append((s, d, il if ol else 0, ol, float(np.float32(rng.rand() * 3))))
    lfin = {S - 1: float(np.float32(rng.rand()))}
    # LM: Q states, arcs reading labels 1..3 or epsilon, writing symbols 10..14 or nothing
    Q = int(rng.randint(2, 5))
    lm = []
    for q in range(Q):
        for _ in range(int(rng.randint(1, 5))):
            il = int(rng.randint(0, 4))
            d = int(rng.randint(0, Q))
            if il == 0 and d <= q:
                d = (q + 1) % Q if q + 1 < Q else q     # epsilon arcs only forward: no epsilon cycles
                if d == q:
                    continue
            lm.append((q, d, il, int(rng.choice([0, 10, 11, 12, 13, 14])), float(np.float32(rng.rand() * 2))))
    mfin = {int(q): float(np.float32(rng.rand())) for q in rng.choice(Q, size=int(rng.randint(1, Q + 1)), replace=False)}
    lf, mf = str(tmp_path / "lat.txt"), str(tmp_path / "lm.txt")
    write_fst(lf, lat, lfin, 0)
    write_fst(mf, lm, mfin, 0)
    if not any(a[0] == 0 for a in lm):
        pytest.skip("LM start state has no arcs")
    r = subprocess.run([exe, lf, mf], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr
    ref = brute(lat, lfin, 0, lm, mfin, 0, max_eps=Q)
    if not ref:
        assert r.stdout.strip() == "nopath"
        return
    lines = r.stdout.strip().split("\n")
    total = float(lines[0].split()[1])
    assert abs(total - ref[0][0]) < 1e-5 * max(1.0, abs(ref[0][0])), (total, ref[0])
    chain = [l.split() for l in lines[1:-1]]
    # the chain's weights (label-free arcs folded in) add up to the total
    assert abs(sum(float(c[2]) for c in chain) + float(lines[-1].split()[1]) - total) < 1e-4
    if len(ref) == 1 or ref[1][0] - ref[0][0] > 1e-4:      # unique optimum: labels must agree
        assert [int(c[0]) for c in chain if int(c[0])] == ref[0][1]
        assert [int(c[1]) for c in chain if int(c[1])] == ref[0][2]


@pytest.fixture(scope="module")
def exe_chain(tmp_path_factory):
    lib = os.path.join(ROOT, "asr-craft_amd", "lib")
    out = str(tmp_path_factory.mktemp("compose") / "compose_chain")
    r = subprocess.run(["g++", "-O1", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "asr-craft_amd", "host"),
                        os.path.join(ROOT, "tests", "host", "compose_chain.cpp"), "-o", out, "-L" + lib, "-Wl,-rpath," + lib,
                        "-lcrf_amd_host", "-lscrf_amd"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    return out


def machine_walks(arcs, fin, start, labels, max_eps):
    """every accepting walk of a transducer over the input string `labels`: yields (cost, output labels)"""
    out = {}
    for a in arcs:
        out.setdefault(a[0], []).append(a)

    def rec(q, k, cost, outs, eps_run):
        if k == len(labels) and q in fin:
            yield cost + fin[q], outs
        for a in out.get(q, []):
            if a[2] == 0 and eps_run < max_eps:
                yield from rec(a[1], k, cost + a[4], outs + ([a[3]] if a[3] else []), eps_run + 1)
            elif k < len(labels) and a[2] == labels[k] and a[2] != 0:
                yield from rec(a[1], k + 1, cost + a[4], outs + ([a[3]] if a[3] else []), 0)
    yield from rec(start, 0, 0.0, [], 0)


@pytest.mark.parametrize("seed", range(10))
def test_dictionary_then_lm_chain_equals_exhaustive_enumeration(exe_chain, tmp_path, seed):
    """lattice o (dict o LM), CRFFstDecode's chain (Main.cpp:929-1006): composeFst for the two static machines, then
    the product search -- against every (lattice path, dictionary walk, LM walk) triple."""
    rng = np.random.RandomState(4200 + seed)
    S = int(rng.randint(4, 7))
    lat = []
    for s in range(S - 1):
        for _ in range(int(rng.randint(1, 4))):
            d = int(rng.randint(s + 1, min(S, s + 3)))
            il = int(rng.randint(1, 4))
            lat.append((s, d, il, il, float(np.float32(rng.rand() * 3))))
    lfin = {S - 1: float(np.float32(rng.rand()))}
    # dictionary: phones 1..3 in, words 20..22 out (on some arcs), epsilon-input arcs only forward
    Qd = int(rng.randint(2, 4))
    dic = []
    for q in range(Qd):
        for _ in range(int(rng.randint(2, 5))):
            il = int(rng.randint(0, 4))
            d = int(rng.randint(0, Qd))
            if il == 0:
                if q + 1 >= Qd:
                    continue
                d = q + 1
            dic.append((q, d, il, int(rng.choice([0, 0, 20, 21, 22])), float(np.float32(rng.rand()))))
    dfin = {int(q): float(np.float32(rng.rand())) for q in rng.choice(Qd, size=int(rng.randint(1, Qd + 1)), replace=False)}
    # LM over the words, writing them through
    Ql = int(rng.randint(1, 4))
    lm = []
    for q in range(Ql):
        for w in (20, 21, 22):
            if rng.rand() < 0.75:
                lm.append((q, int(rng.randint(0, Ql)), w, w, float(np.float32(rng.rand() * 2))))
        if q + 1 < Ql and rng.rand() < 0.5:
            lm.append((q, q + 1, 0, 0, float(np.float32(rng.rand()))))
    mfin = {int(q): float(np.float32(rng.rand())) for q in rng.choice(Ql, size=int(rng.randint(1, Ql + 1)), replace=False)}
    if not any(a[0] == 0 for a in dic) or not any(a[0] == 0 for a in lm):
        pytest.skip("a start state has no arcs")
    lf, df, mf = str(tmp_path / "lat.txt"), str(tmp_path / "dict.txt"), str(tmp_path / "lm.txt")
    write_fst(lf, lat, lfin, 0)
    write_fst(df, dic, dfin, 0)
    write_fst(mf, lm, mfin, 0)
    r = subprocess.run([exe_chain, "chain", lf, df, mf], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr
    # brute force
    lout = {}
    for a in lat:
        lout.setdefault(a[0], []).append(a)
    res = []

    def lat_paths(s, cost, ols):
        if s in lfin:
            for c1, words in machine_walks(dic, dfin, 0, ols, Qd):
                for c2, outs in machine_walks(lm, mfin, 0, words, Ql):
                    res.append((cost + lfin[s] + c1 + c2, list(ols), outs))
        for a in lout.get(s, []):
            lat_paths(a[1], cost + a[4], ols + [a[3]])
    lat_paths(0, 0.0, [])
    res.sort(key=lambda t: t[0])
    lines = r.stdout.strip().split("\n")
    assert lines[0].startswith("states ")
    if not res:
        assert lines[1] == "nopath"
        return
    total = float(lines[1].split()[1])
    assert abs(total - res[0][0]) < 1e-5 * max(1.0, abs(res[0][0])), (total, res[0])
    chain = [l.split() for l in lines[2:-1]]
    if len(res) == 1 or res[1][0] - res[0][0] > 1e-4:
        assert [int(c[0]) for c in chain if int(c[0])] == res[0][1]
        assert [int(c[1]) for c in chain if int(c[1])] == res[0][2]


def test_mlf_manager_keys_symbols_and_acceptor(exe_chain, tmp_path):
    """io/CRF_MLFManager.cpp: key = text between the last '/' and the last '.', one symbol per line looked up whole
    (-1 when missing), entries closed by a single '.', getFst = linear acceptor id:id/0."""
    (tmp_path / "sym.txt").write_text("<eps> 0\nsil 1\nhello 2\nworld 3\n")
    (tmp_path / "a.mlf").write_text('#!MLF!#\n"*/utt1.lab"\nsil\nhello\n\nworld\n.\n"dir/sub/utt2.rec"\nworld\nunknown word\n.\n')
    r = subprocess.run([exe_chain, "mlf", str(tmp_path / "a.mlf"), str(tmp_path / "sym.txt"), "somewhere/utt1.htk"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr
    assert r.stdout.split("\n")[:5] == ["start 0 states 4", "0 1 1 1 0", "1 2 2 2 0", "2 3 3 3 0", "final 3 0"]
    r = subprocess.run([exe_chain, "mlf", str(tmp_path / "a.mlf"), str(tmp_path / "sym.txt"), "utt2.x"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr
    assert r.stdout.split("\n")[:4] == ["start 0 states 3", "0 1 3 3 0", "1 2 -1 -1 0", "final 2 0"]
    r = subprocess.run([exe_chain, "mlf", str(tmp_path / "a.mlf"), str(tmp_path / "sym.txt"), "utt3.lab"], capture_output=True, text=True, timeout=60)
    assert r.returncode != 0 and "no MLF entry" in r.stderr
    (tmp_path / "bad.mlf").write_text('"x.lab"\nsil\n.\n')
    r = subprocess.run([exe_chain, "mlf", str(tmp_path / "bad.mlf"), str(tmp_path / "sym.txt"), "x.lab"], capture_output=True, text=True, timeout=60)
    assert r.returncode != 0 and "not a wellformed MLF" in r.stderr
