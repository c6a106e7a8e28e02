"""not-gpu: the FST operations behind CRFFstDecode's phone-penalty and pruning stages (CRFFstDecode/src/Main.cpp:896-940)
-- crf_amd::composeFst with the sequencing epsilon filter, rmEpsilonLog (RmEpsilon between StdToLogMapper and
LogToStdMapper), pruneFst, topSortFst (libcrf_amd_host.so) -- against exhaustive path enumeration on random small
acyclic machines.  OpenFST is not in the tree; what is checked is the published semantics of each operation:
  compose + filter : every (path of a, path of b) pair that matches is ONE path of the result, weight = the sum
  rmEpsilonLog     : no epsilon:epsilon arc is left; per (input string, output string) the LOG-semiring sum over all
                     paths is what it was; where epsilon paths fan into one labelled arc the arc carries their log-sum
  prune            : every successful path of weight <= best + threshold survives, every arc left lies on one
  topsort          : src < dst on every arc, same paths"""
import math
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
    out = str(tmp_path_factory.mktemp("fstops") / "fst_ops")
    r = subprocess.run(["g++", "-O1", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "asr-craft_amd", "host"),
                        os.path.join(ROOT, "tests", "host", "fst_ops.cpp"), "-o", out, "-L" + lib, "-Wl,-rpath," + lib,
                        "-lcrf_amd_host", "-lscrf_amd"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    return out


def write_fst(path, arcs, finals, start):
    with open(path, "w") as f:
        arcs = sorted(arcs, key=lambda a: a[0] != start)      # text format: the first line's source is the start state
        if not arcs:
            f.write("%d %.9g\n" % (start, finals.get(start, 0.0)))
        for a in arcs:
            f.write("%d %d %d %d %.9g\n" % a)
        for s, w in finals.items():
            f.write("%d %.9g\n" % (s, w))


def run(exe, *args):
    r = subprocess.run([exe] + [str(a) for a in args], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    return parse(r.stdout)


def parse(text):
    arcs, finals, start, n = [], {}, None, 0
    for ln in text.splitlines():
        t = ln.split()
        if not t:
            continue
        if t[0] == "cyclic":
            return None
        if t[0] == "start":
            start, n = int(t[1]), int(t[3])
        elif t[0] == "final":
            finals[int(t[1])] = float(t[2])
        else:
            arcs.append((int(t[0]), int(t[1]), int(t[2]), int(t[3]), float(t[4])))
    return arcs, finals, start, n


def paths(arcs, finals, start, limit=200000):
    """every successful path of an acyclic machine: [(weight, ilabels with eps, olabels with eps, arc indices)]"""
    out = {}
    for i, a in enumerate(arcs):
        out.setdefault(a[0], []).append((i, a))
    res = []

    def go(s, w, il, ol, ix):
        assert len(res) < limit
        if s in finals:
            res.append((w + finals[s], il, ol, ix))
        for i, a in out.get(s, []):
            go(a[1], w + a[4], il + (a[2],), ol + (a[3],), ix + (i,))
    if start is not None:
        go(start, 0.0, (), (), ())
    return res


def strip(seq):
    return tuple(x for x in seq if x)


def logadd(ws):
    m = min(ws)
    return m - math.log(sum(math.exp(-(w - m)) for w in ws))


def random_acyclic(rng, S, n_lab, p_eps_in, p_eps_out, n_final=1, max_out=3, wscale=3.0):
    arcs = []
    for s in range(S - 1):
        for _ in range(int(rng.randint(1, max_out + 1))):
            d = int(rng.randint(s + 1, min(S, s + 3)))
            il = 0 if rng.rand() < p_eps_in else int(rng.randint(1, n_lab + 1))
            ol = 0 if rng.rand() < p_eps_out else int(rng.randint(1, n_lab + 1))
            arcs.append((s, d, il, ol, float(np.float32(rng.rand() * wscale - 0.5))))
    finals = {S - 1: float(np.float32(rng.rand()))}
    for _ in range(n_final - 1):
        finals[int(rng.randint(1, S))] = float(np.float32(rng.rand() * 2))
    return arcs, finals, 0


@pytest.mark.parametrize("seed", range(10))
def test_compose_with_the_sequencing_filter_holds_every_matching_pair_of_paths_once(exe, tmp_path, seed):
    rng = np.random.RandomState(4100 + seed)
    A = random_acyclic(rng, int(rng.randint(3, 6)), 2, 0.0, 0.35, n_final=2)
    B = random_acyclic(rng, int(rng.randint(3, 6)), 2, 0.35, 0.3, n_final=2)
    fa, fb = str(tmp_path / "a.txt"), str(tmp_path / "b.txt")
    write_fst(fa, *A)
    write_fst(fb, *B)
    pa, pb = paths(*A), paths(*B)
    # the pairs: a's output string == b's input string; identified by (arc indices of a, arc indices of b)
    want = sorted(round(wa + wb, 4) for wa, _, oa, _ in pa for wb, ib, _, _ in pb if strip(oa) == strip(ib))
    got_f = run(exe, "compose", fa, fb, 1)
    got = sorted(round(p[0], 4) for p in paths(*got_f[:3]))
    assert len(got) == len(want)
    assert np.allclose(got, want, atol=2e-4)
    # without the filter the same best weight (the epsilon interleavings only add copies)
    got_n = run(exe, "compose", fa, fb, 0)
    pn = [p[0] for p in paths(*got_n[:3])]
    assert len(pn) >= len(want)
    if want:
        assert abs(min(pn) - want[0]) < 2e-4


@pytest.mark.parametrize("seed", range(12))
def test_rm_epsilon_on_the_log_semiring_keeps_the_log_sum_of_every_string_pair(exe, tmp_path, seed):
    rng = np.random.RandomState(5200 + seed)
    S = int(rng.randint(4, 8))
    arcs, finals, start = random_acyclic(rng, S, 2, 0.0, 0.0, n_final=2, max_out=2)
    # epsilon:epsilon arcs, in parallel pairs and chains
    for _ in range(int(rng.randint(2, 6))):
        s = int(rng.randint(0, S - 1))
        d = int(rng.randint(s + 1, min(S, s + 3)))
        arcs.append((s, d, 0, 0, float(np.float32(rng.rand() * 2))))
    # some arcs with epsilon on one side only (they stay)
    arcs = [(a[0], a[1], a[2], 0 if (a[2] and rng.rand() < 0.2) else a[3], a[4]) for a in arcs]
    f = str(tmp_path / "m.txt")
    write_fst(f, arcs, finals, start)
    before = {}
    for w, il, ol, _ in paths(arcs, finals, start):
        before.setdefault((strip(il), strip(ol)), []).append(w)
    got = run(exe, "rmeps", f)
    assert all(not (a[2] == 0 and a[3] == 0) for a in got[0])
    # merged: no two arcs of a state share (ilabel, olabel, next state)
    keys = [(a[0], a[1], a[2], a[3]) for a in got[0]]
    assert len(keys) == len(set(keys))
    after = {}
    for w, il, ol, _ in paths(*got[:3]):
        after.setdefault((strip(il), strip(ol)), []).append(w)
    assert set(after) == set(before)
    for k in before:
        assert abs(logadd(after[k]) - logadd(before[k])) < 1e-4 * max(1.0, abs(logadd(before[k]))), (k, after[k], before[k])


def test_rm_epsilon_log_sums_parallel_epsilon_paths_into_the_arc(exe, tmp_path):
    """two epsilon paths 0 -> 1 (weights 0.5 and 1.25) in front of one labelled arc 1 -> 2 (weight 2): the arc that is
    left weighs -log(e^-0.5 + e^-1.25) + 2, not min(0.5, 1.25) + 2 -- what separates the log from the tropical removal"""
    f = str(tmp_path / "m.txt")
    write_fst(f, [(0, 1, 0, 0, 0.5), (0, 3, 0, 0, 0.25), (3, 1, 0, 0, 1.0), (1, 2, 7, 8, 2.0)], {2: 0.125}, 0)
    arcs, finals, start, n = run(exe, "rmeps", f)
    assert n == 2 and len(arcs) == 1 and arcs[0][2:4] == (7, 8) and arcs[0][0] == start
    assert abs(arcs[0][4] - (-math.log(math.exp(-0.5) + math.exp(-1.25)) + 2.0)) < 1e-6
    assert list(finals.values()) == [0.125]
    # an epsilon cycle is refused
    write_fst(f, [(0, 1, 0, 0, 0.5), (1, 0, 0, 0, 0.5), (1, 2, 1, 1, 1.0)], {2: 0.0}, 0)
    r = subprocess.run([exe, "rmeps", f], capture_output=True, text=True, timeout=60)
    assert r.returncode != 0 and "epsilon cycle" in r.stderr


@pytest.mark.parametrize("seed", range(10))
def test_prune_keeps_the_arcs_of_the_paths_within_the_threshold(exe, tmp_path, seed):
    rng = np.random.RandomState(6300 + seed)
    M = random_acyclic(rng, int(rng.randint(4, 8)), 3, 0.1, 0.1, n_final=2)
    f = str(tmp_path / "m.txt")
    write_fst(f, *M)
    allp = sorted(p[0] for p in paths(*M))
    thr = float(np.float32(rng.rand() * 2.5))
    # a threshold that sits within float rounding of a path weight would make the comparison a coin toss: move it
    while any(abs((allp[0] + thr) - w) < 1e-4 for w in allp):
        thr += 3e-4
    got = run(exe, "prune", f, "%.9g" % thr)
    kept = paths(*got[:3])
    want = [w for w in allp if w <= allp[0] + thr]
    # every path within the threshold is still there ...
    kw = sorted(p[0] for p in kept)
    for w in want:
        assert any(abs(w - k) < 2e-4 for k in kw)
    assert abs(kw[0] - allp[0]) < 2e-4
    # ... and every arc and final state left lies on one of them (Prune works on arcs: arcs of two different good
    # paths may still combine into a path beyond the threshold)
    good_arcs, good_finals = set(), set()
    for p in kept:
        if p[0] <= allp[0] + thr + 1e-4:
            good_arcs.update(p[3])
            s_ = got[2]
            for i in p[3]:
                s_ = got[0][i][1]
            good_finals.add(s_)
    assert good_arcs == set(range(len(got[0])))
    assert good_finals == set(got[1])
    assert len(kept) <= len(allp)


@pytest.mark.parametrize("seed", range(6))
def test_topsort_orders_the_states_and_keeps_the_paths(exe, tmp_path, seed):
    rng = np.random.RandomState(7400 + seed)
    arcs, finals, start = random_acyclic(rng, int(rng.randint(4, 9)), 3, 0.1, 0.1, n_final=2)
    S = 1 + max(max(a[0], a[1]) for a in arcs)
    perm = rng.permutation(S)
    arcs2 = [(int(perm[a[0]]), int(perm[a[1]]), a[2], a[3], a[4]) for a in arcs]
    finals2 = {int(perm[s]): w for s, w in finals.items()}
    f = str(tmp_path / "m.txt")
    write_fst(f, arcs2, finals2, int(perm[start]))
    got = run(exe, "topsort", f)
    assert all(a[0] < a[1] for a in got[0])
    key = lambda p: (round(p[0], 4), p[1], p[2])
    assert sorted(map(key, paths(*got[:3]))) == sorted(map(key, paths(arcs, finals, start)))
    # a cycle is reported, not sorted
    write_fst(f, [(0, 1, 1, 1, 0.0), (1, 2, 1, 1, 0.0), (2, 1, 2, 2, 0.0)], {2: 0.0}, 0)
    assert run(exe, "topsort", f) is None
