"""not-gpu: the split arithmetic of the C++ data-parallel layer (libcrf_amd_host.so: crf_amd_view_range,
crf_amd_minibatch_share -- what CRF_Minibatch_GradAccumulator and the stream manager's children use, and
what every rank of a multi-GPU CRFTrain applies to itself) walked over whole epochs for N = 1, 2, 3 streams
and reduced with the oracle's restatement of the reference's join (sum in stream order / active streams,
trainers/accumulators/CRF_Minibatch_GradAccumulator.cpp:229-312): the result must equal a direct
restatement of accumulateGradient, step for step, including the end-of-epoch flag."""
import ctypes as C
import os

import numpy as np
import pytest

import orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def host():
    p = os.path.join(ROOT, "asr-craft_amd", "lib", "libcrf_amd_host.so")
    if not os.path.exists(p):
        pytest.fail("libcrf_amd_host.so not built: run __graft_entry__.build()")
    lib = C.CDLL(p)
    lib.crf_amd_minibatch_share.restype = C.c_uint32
    lib.crf_amd_minibatch_share.argtypes = [C.c_uint32] * 3
    lib.crf_amd_view_range.argtypes = [C.c_uint32] * 3 + [C.POINTER(C.c_uint32)] * 2
    return lib


def view(lib, U, N, s):
    lo, hi = C.c_uint32(), C.c_uint32()
    lib.crf_amd_view_range(U, N, s, C.byref(lo), C.byref(hi))
    return lo.value, hi.value


@pytest.mark.parametrize("N", [1, 2, 3])
@pytest.mark.parametrize("U,mb", [(7, 3), (10, 4), (5, 5), (9, 100)])
def test_rank_walk_equals_the_reference_accumulator(host, N, U, mb):
    if mb < N:
        pytest.skip("minibatch smaller than the number of streams is refused (setMinibatch)")
    rng = np.random.RandomState(U * 31 + mb)
    n = 6
    ug = rng.normal(size=(U, n))            # stand-in per-utterance gradients
    # views partition the utterances contiguously, last stream takes the remainder (:425-464)
    vs = [view(host, U, N, s) for s in range(N)]
    assert vs[0][0] == 0 and vs[-1][1] == U and all(vs[s][1] == vs[s + 1][0] for s in range(N - 1))
    assert all(vs[s][1] - vs[s][0] == U // N for s in range(N - 1))
    shares = [host.crf_amd_minibatch_share(mb, N, s) for s in range(N)]
    assert sum(shares) == mb and max(shares) - min(shares) <= 1 and shares == sorted(shares, reverse=True)
    # every rank walks its own view; the steps are joined with the oracle's reduce
    pos = [v[0] for v in vs]
    steps = []
    while any(pos[s] < vs[s][1] for s in range(N)):
        sg = np.zeros((N, n)); act = []; cnt = 0
        for s in range(N):
            a = pos[s] < vs[s][1]
            act.append(1 if a else 0)
            k = 0
            while a and pos[s] < vs[s][1] and (k < shares[s] or k == 0):
                sg[s] += ug[pos[s]]; pos[s] += 1; k += 1
            cnt += k
        ended = sum(1 for s in range(N) if pos[s] >= vs[s][1])
        steps.append((orc.minibatch_reduce(sg, act), cnt, ended == N))
    # direct restatement: per step, stream s takes `share` utterances from its view; grad = sum / active
    pos = [v[0] for v in vs]
    i = 0
    while any(pos[s] < vs[s][1] for s in range(N)):
        tot = np.zeros(n); active = 0; cnt = 0
        for s in range(N):
            if pos[s] >= vs[s][1]:
                continue
            active += 1
            take = min(shares[s], vs[s][1] - pos[s])
            tot += ug[pos[s]:pos[s] + take].sum(0)
            pos[s] += take; cnt += take
        g, c, end = steps[i]
        np.testing.assert_allclose(g, tot / active, rtol=1e-13, atol=1e-15)
        assert c == cnt and end == all(pos[s] >= vs[s][1] for s in range(N))
        i += 1
    assert i == len(steps) and steps[-1][2]


def test_whole_file_minibatch_share(host):
    assert host.crf_amd_minibatch_share(0xFFFFFFFF, 4, 2) == 0xFFFFFFFF     # crf_bunch_size=0: the whole view
