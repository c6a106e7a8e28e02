"""-m "not gpu": the N > 1 control flow of the native trainer, executed as N real processes without a GPU.

bin/CRFTrain's sources (asr-craft_amd/host/CRFTrain_main.cpp + crf_amd.cpp) are linked against tests/host/scrf_stub.cpp,
a TEST-ONLY stand-in for the C ABI: canned per-utterance gradients, the real optimiser arithmetic, and a file-based
"collective" that sums in rank order and divides by the active ranks like scrf_allreduce_grad_ex.  What is under test is
everything ABOVE the ABI in one-process-per-GPU mode -- rank r = the reference's stream r over the contiguous view
[r floor(U/N), ...) (io/CRF_FeatureStreamManager.cpp:425-464), the minibatch share floor(mb/N) + (r < mb mod N)
(trainers/accumulators/CRF_Minibatch_GradAccumulator.cpp:229-241), sum / active streams, a rank whose view is exhausted,
the epoch ending when every stream is exhausted (:296-312), rank 0 as the only writer, the communicator-id handshake,
and that a rank that fails takes the others down with it instead of leaving them in the collective."""
import os
import subprocess
import time

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "asr-craft_amd", "host")
L, W = 6, 3


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("stubbin") / "CRFTrain_stub")
    subprocess.run(["g++", "-O1", "-ffp-contract=off", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-I" + HOST,
                    os.path.join(HOST, "CRFTrain_main.cpp"), os.path.join(HOST, "crf_amd.cpp"),
                    os.path.join(ROOT, "tests", "host", "scrf_stub.cpp"), "-o", out], check=True, timeout=600)
    return out


@pytest.fixture(scope="module")
def data(tmp_path_factory):
    """11 utterances of 2..6 frames, 3 features in eighths (exact in float), 6 labels, as ascii files."""
    d = tmp_path_factory.mktemp("mrdata")
    rng = np.random.RandomState(5)
    utts = []
    with open(d / "f.ascii", "w") as ff, open(d / "l.ascii", "w") as lf:
        for u in range(11):
            T = int(rng.randint(2, 7))
            X = rng.randint(0, 17, size=(T, W)) / 8.0
            lab = rng.randint(0, L, size=T)
            for t in range(T):
                ff.write("%d %d %s\n" % (u, t, " ".join("%g" % v for v in X[t])))
                lf.write("%d %d %d\n" % (u, t, lab[t]))
            utts.append((X.astype(np.float32), lab))
    return str(d), utts


def flags(d, out, **kw):
    f = dict(crf_epochs=3, crf_lr=0.1, crf_bunch_size=3, threads=1, crf_utt_rpt=1, crf_train_order="seq", crf_featuremap="stdstate")
    f.update(kw)
    return ["ftr1_file=" + os.path.join(d, "f.ascii"), "ftr1_format=ascii", "hardtarget_file=" + os.path.join(d, "l.ascii"),
            "crf_label_size=%d" % L, "crf_model_type=stdframe", "label_maximum_duration=1",
            "out_weight_file=" + out] + ["%s=%s" % kv for kv in f.items()]


def launch(exe, d, outdir, world, delay=None, env_extra=None, **kw):
    """one process per rank, like the launcher; returns [(rc, stdout, stderr)] by rank"""
    os.makedirs(outdir, exist_ok=True)
    comm = os.path.join(outdir, "comm")
    os.makedirs(comm, exist_ok=True)
    procs = []
    for r in range(world):
        e = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), SCRF_STUB_COMM_DIR=comm,
                 SCRF_COMM_TIMEOUT_S="30", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
        e.update(env_extra or {})
        if delay and r in delay:
            time.sleep(delay[r])
        procs.append(subprocess.Popen([exe] + flags(d, os.path.join(outdir, "w.out"), **kw), env=e,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    res = []
    for p in procs:
        try:
            o, er = p.communicate(timeout=120)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        res.append((p.returncode, o, er))
    return res


def weight_files(outdir):
    return {n: open(os.path.join(outdir, n), "rb").read() for n in sorted(os.listdir(outdir)) if n.startswith("w.out")}


def replay(utts, N, mb, epochs, lr=0.1):
    """the reference's minibatch protocol over the stub's canned gradients; returns (lambda, lambdaAvg)"""
    nsf, ntf = W + 1, 1
    n = L * (nsf + L * ntf)
    idx = np.arange(n, dtype=np.uint64)
    keys = [np.uint64(X.shape[0] * 31 + int(lab.sum()) + int(np.trunc(X.astype(np.float32) * np.float32(16)).sum())) for X, lab in utts]
    U = len(utts)
    per = U // N
    views = [(s * per, U if s == N - 1 else (s + 1) * per) for s in range(N)]
    lam = np.zeros(n); acc = np.zeros(n); avg = np.zeros(n)
    acc_cnt = 0
    lr = float(np.float32(lr))
    for _ in range(epochs):
        pos = [lo for lo, _ in views]
        assert any(pos[s] < views[s][1] for s in range(N))
        while any(pos[s] < views[s][1] for s in range(N)):
            g = np.zeros(n); active = 0; inc = 0
            for s in range(N):
                if pos[s] >= views[s][1]:
                    continue
                share = mb // N + (1 if s < mb % N else 0)
                take = min(max(share, 1), views[s][1] - pos[s])
                for u in range(pos[s], pos[s] + take):
                    g += ((keys[u] + np.uint64(13) * idx) % np.uint64(97)).astype(np.float64) / 64.0 - 48.0 / 64.0 - np.floor(lam * 8.0) / 1024.0
                pos[s] += take; active += 1; inc += take
            g = g / active
            lam = lam + lr * g
            acc = acc + lam
            acc_cnt += inc
        avg = acc / float(np.float32(acc_cnt))
    return lam, avg


def as_file(v):
    return ("".join("%g\n" % x for x in v)).encode()


@pytest.mark.parametrize("world,bunch", [(2, 3), (3, 4), (4, 5)])
def test_n_ranks_equal_one_process_with_n_streams_and_the_replayed_protocol(exe, data, tmp_path, world, bunch):
    d, utts = data
    res = launch(exe, d, str(tmp_path / "dist"), world, crf_bunch_size=bunch)
    for r, (rc, o, er) in enumerate(res):
        assert rc == 0, "rank %d: %s %s" % (r, o[-400:], er[-400:])
    # rank 0 is the only writer of files and progress lines
    assert "Writing Final Iteration weights" in res[0][1]
    for r in range(1, world):
        assert "Writing" not in res[r][1] and "Iteration:" not in res[r][1]
    dist = weight_files(str(tmp_path / "dist"))
    assert "w.out.rccl_id" not in dist and len(dist) == 2 + 2 * 3     # final + avg, per-epoch pairs; the id file is gone
    assert os.path.exists(str(tmp_path / "dist" / ".done.train"))
    # one process, N streams (the path the GPU tests pin against the oracle) writes the same bytes
    os.makedirs(str(tmp_path / "one"))
    r1 = subprocess.run([exe] + flags(d, str(tmp_path / "one" / "w.out"), crf_bunch_size=bunch, threads=world),
                        capture_output=True, text=True, timeout=120)
    assert r1.returncode == 0, r1.stderr
    one = weight_files(str(tmp_path / "one"))
    assert sorted(one) == sorted(dist) and all(one[k] == dist[k] for k in one)
    # and both equal the protocol replayed in Python: share split, / active streams, views exhausted at different
    # times (world = 4: 11 utterances -> views of 2, 2, 2 and 5), epoch end when all are
    lam, avg = replay(utts, world, bunch, 3)
    assert dist["w.out"] == as_file(lam)
    assert dist["w.out.avg.out"] == as_file(avg)


@pytest.mark.parametrize("world,bunch", [(2, 3), (4, 5)])
def test_two_block_collective_with_transition_features(exe, data, tmp_path, world, bunch):
    """With transition features the per-step collective runs in two blocks (transition weights, then state weights +
    scalars; the engine issues the first under the state contraction inside scrf_fb_batch_allreduce).  The block
    sequence is part of the protocol: ranks that go through the fused call and ranks whose view is exhausted (plain
    scrf_allreduce_grad_ex, world = 4 has views of 2, 2, 2 and 5 utterances) must still meet, and N processes must
    write the bytes one process with N streams writes."""
    d, utts = data
    res = launch(exe, d, str(tmp_path / "dist"), world, crf_bunch_size=bunch, crf_featuremap="stdtrans")
    for r, (rc, o, er) in enumerate(res):
        assert rc == 0, "rank %d: %s %s" % (r, o[-400:], er[-400:])
    dist = weight_files(str(tmp_path / "dist"))
    # two rounds of the file collective per step: more rounds than steps
    rounds = len([n for n in os.listdir(os.path.join(str(tmp_path / "dist"), "comm", os.listdir(os.path.join(str(tmp_path / "dist"), "comm"))[0]))
                  if n.startswith("r") and n.endswith(".0")])
    os.makedirs(str(tmp_path / "one"))
    r1 = subprocess.run([exe] + flags(d, str(tmp_path / "one" / "w.out"), crf_bunch_size=bunch, threads=world, crf_featuremap="stdtrans"),
                        capture_output=True, text=True, timeout=120)
    assert r1.returncode == 0, r1.stderr
    one = weight_files(str(tmp_path / "one"))
    assert sorted(one) == sorted(dist) and all(one[k] == dist[k] for k in one)
    assert rounds % 2 == 0 and rounds >= 2 * 3     # two blocks per step, at least one step per epoch


def test_more_ranks_than_utterances_leaves_empty_views_inactive(exe, tmp_path):
    d = tmp_path / "tiny"
    d.mkdir()
    with open(d / "f.ascii", "w") as ff, open(d / "l.ascii", "w") as lf:
        for u in range(2):
            for t in range(3):
                ff.write("%d %d %g %g %g\n" % (u, t, 0.125 * (u + t), 0.5, 1.0))
                lf.write("%d %d %d\n" % (u, t, (u + t) % L))
    res = launch(exe, str(d), str(tmp_path / "o3"), 3, crf_bunch_size=3, crf_epochs=2)
    assert all(rc == 0 for rc, _, _ in res), [er[-300:] for _, _, er in res]
    utts = [(np.array([[0.125 * (u + t), 0.5, 1.0] for t in range(3)], dtype=np.float32), np.array([(u + t) % L for t in range(3)])) for u in range(2)]
    lam, _ = replay(utts, 3, 3, 2)        # views: [0,0) [0,0) [0,2): two ranks never have an utterance
    assert weight_files(str(tmp_path / "o3"))["w.out"] == as_file(lam)


def test_a_failing_rank_stops_every_rank(exe, data, tmp_path):
    d, _ = data
    t0 = time.time()
    res = launch(exe, d, str(tmp_path / "fail"), 3, crf_bunch_size=3, env_extra={"SCRF_STUB_FAIL_RANK": "1", "SCRF_STUB_FAIL_AT": "2"})
    assert time.time() - t0 < 60                       # nobody waited for a timeout
    assert all(rc != 0 for rc, _, _ in res)
    assert "injected numeric failure" in res[1][2]
    assert "other rank(s) failed" in res[0][2] and "other rank(s) failed" in res[2][2]
    assert not os.path.exists(str(tmp_path / "fail" / "w.out"))


def test_a_rank_whose_collective_itself_fails_aborts_and_the_peers_stop(exe, data, tmp_path):
    """The failure-flag protocol covers a rank that fails in its SHARE.  A rank whose all-reduce call itself errors (a
    device error around the collective) cannot send a flag: it aborts the communicator and exits non-zero, and its
    peers' bounded wait ends with the asynchronous error -- nobody hangs, nobody writes a weight file."""
    d, _ = data
    t0 = time.time()
    res = launch(exe, d, str(tmp_path / "cfail"), 3, crf_bunch_size=3,
                 env_extra={"SCRF_STUB_COLL_FAIL_RANK": "1", "SCRF_STUB_COLL_FAIL_AT": "2", "SCRF_COMM_TIMEOUT_S": "30"})
    assert time.time() - t0 < 25                       # the abort ended the peers' wait, not the 30 s watchdog
    assert all(rc != 0 for rc, _, _ in res)
    assert "injected failure inside the collective" in res[1][2] and "communicator aborted" in res[1][2]
    assert "aborted the communicator" in res[0][2] and "aborted the communicator" in res[2][2]
    assert not os.path.exists(str(tmp_path / "cfail" / "w.out"))


def test_a_rank_that_dies_inside_the_collective_is_found_by_the_watchdog(exe, data, tmp_path):
    """a killed rank sends nothing at all: the peers' wait is bounded by SCRF_COMM_TIMEOUT_S and ends non-zero"""
    d, _ = data
    t0 = time.time()
    res = launch(exe, d, str(tmp_path / "cdie"), 2, crf_bunch_size=2,
                 env_extra={"SCRF_STUB_COLL_FAIL_RANK": "1", "SCRF_STUB_COLL_FAIL_AT": "2", "SCRF_STUB_COLL_DIE": "1", "SCRF_COMM_TIMEOUT_S": "3"})
    assert 2.5 < time.time() - t0 < 40
    assert res[1][0] != 0 and res[0][0] != 0
    assert "did not complete within SCRF_COMM_TIMEOUT_S" in res[0][2]
    assert not os.path.exists(str(tmp_path / "cdie" / "w.out"))


def test_stale_id_file_and_a_late_rank(exe, data, tmp_path):
    """a leftover id file of another launch (other token) is ignored, and a rank that starts seconds after rank 0
    published the id still joins -- the handshake compares launch tokens, not clocks"""
    d, _ = data
    out = tmp_path / "late"
    out.mkdir()
    with open(out / "w.out.rccl_id", "wb") as f:
        f.write(b"\x07" * 128 + b"launch:127.0.0.1:29999:-:-:ppid1")
    res = launch(exe, d, str(out), 2, delay={1: 3.0}, crf_bunch_size=2)
    assert all(rc == 0 for rc, _, _ in res), [er[-300:] for _, _, er in res]
    assert not os.path.exists(str(out / "w.out.rccl_id"))
    # ranks that do not share a launch token never meet: the one without rank 0's token times out, and says so
    out2 = tmp_path / "mismatch"
    out2.mkdir()
    os.makedirs(str(out2 / "comm"))
    procs = []
    for r, tok in [(0, "a"), (1, "b")]:
        e = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), SCRF_STUB_COMM_DIR=str(out2 / "comm"),
                 SCRF_COMM_TIMEOUT_S="3", SCRF_LAUNCH_TOKEN=tok)
        procs.append(subprocess.Popen([exe] + flags(d, str(out2 / "w.out")), env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    o1, e1 = procs[1].communicate(timeout=60)
    procs[0].kill()
    procs[0].communicate()
    assert procs[1].returncode != 0 and "timed out waiting for rank 0's communicator id file" in e1
