"""N>1 path on CPU: world_size-2 (and 3) gloo processes shard a minibatch exactly like the
reference's stream threads; gradients come from the CPU oracle, the collective from
scrf_amd.dist (the same helper bench.py uses over RCCL).  Checks against the oracle's
single-process restatement of CRF_Minibatch_GradAccumulator::accumulateGradient."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _problem():
    sys.path.insert(0, HERE); sys.path.insert(0, os.path.join(ROOT, "asr-craft_amd", "python"))
    import orc
    from scrf_amd import synth
    L, D, in_w, U, T = 4, 3, 3, 7, 10
    frames, labels, off = synth.make_batch(U, T, in_w, L, D, seed=5, t_jitter=3)
    cfg = orc.config(L=L, D=D, F=8 * in_w + D)
    lay = orc.Layout(cfg)
    lam = synth.make_lambda(lay.lambda_len, scale=0.2)
    return orc, cfg, lay, lam, frames, labels, off, U, D


def _utt_grad(orc, cfg, lay, lam, frames, labels, off, D, u, grad):
    a, b = int(off[u]), int(off[u + 1])
    X = orc.windows(frames[a:b], D)
    rc, grad, n, z = orc.seg_build_gradient(cfg, lay, lam, X, labels[a:b], b - a, grad=grad)
    assert rc == 0
    return n, z


def _worker(rank, world, port, minibatch, out_dir, one_collective=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    orc, cfg, lay, lam, frames, labels, off, U, D = _problem()
    from scrf_amd.dist import MinibatchReducer, RankCursor, reduce_minibatch
    cur = RankCursor(U, world, rank)
    red = MinibatchReducer(lay.lambda_len, "cpu") if one_collective else None
    steps = []
    while True:
        active = cur.active
        g = np.zeros(lay.lambda_len); numer = zx = 0.0; n = 0
        for u in cur.next_step(minibatch):
            a, b = _utt_grad(orc, cfg, lay, lam, frames, labels, off, D, u, g)
            numer += a; zx += b; n += 1
        if red is not None:
            # what bench.py runs: gradient and scalars in one buffer, one all-reduce per step
            red.grad.copy_(torch.from_numpy(g))
            red.tail[:3] = torch.tensor([numer, zx, float(n)], dtype=torch.float64)
            red.set_active(active)
            n_active = int(red.reduce().item())
            gt, sc = red.grad.clone(), red.tail[:3].clone()
        else:
            gt = torch.from_numpy(g); sc = torch.tensor([numer, zx, float(n)], dtype=torch.float64)
            n_active = int(reduce_minibatch(gt, sc, active).item())
        if n_active == 0:
            break
        steps.append((gt.numpy().copy(), sc.numpy().copy(), n_active))
    if rank == 0:
        np.save(os.path.join(out_dir, "steps.npy"), np.array([np.concatenate([s[0], s[1], [s[2]]]) for s in steps]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,minibatch,one_collective", [(2, 3, False), (3, 4, False), (2, 2, False), (2, 3, True), (3, 5, True)])
def test_sharded_minibatches_match_accumulator_semantics(tmp_path, world, minibatch, one_collective):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, minibatch, str(tmp_path), one_collective), nprocs=world, join=True)
    got = np.load(os.path.join(str(tmp_path), "steps.npy"))
    # single-process restatement: streams with contiguous views, sum in stream order / n_active
    orc, cfg, lay, lam, frames, labels, off, U, D = _problem()
    sys.path.insert(0, os.path.join(ROOT, "asr-craft_amd", "python"))
    from scrf_amd.dist import RankCursor
    curs = [RankCursor(U, world, r) for r in range(world)]
    k = 0
    total_utts = 0
    while any(c.active for c in curs):
        sg = np.zeros((world, lay.lambda_len)); act = []; numer = zx = 0.0; n = 0
        for r, c in enumerate(curs):
            act.append(1 if c.active else 0)
            for u in c.next_step(minibatch):
                a, b = _utt_grad(orc, cfg, lay, lam, frames, labels, off, D, u, sg[r])
                numer += a; zx += b; n += 1
        exp = orc.minibatch_reduce(sg, act)
        g = got[k, :lay.lambda_len]
        np.testing.assert_allclose(g, exp, rtol=1e-12, atol=1e-13)
        assert abs(got[k, lay.lambda_len] - numer) < 1e-9 and abs(got[k, lay.lambda_len + 1] - zx) < 1e-9
        assert got[k, lay.lambda_len + 2] == n and got[k, -1] == sum(act)
        total_utts += n
        k += 1
    assert k == got.shape[0] and total_utts == U


def test_view_ranges_follow_the_stream_manager():
    sys.path.insert(0, os.path.join(ROOT, "asr-craft_amd", "python"))
    from scrf_amd.dist import minibatch_share, view_range
    assert [view_range(10, 3, r) for r in range(3)] == [(0, 3), (3, 6), (6, 10)]
    assert [minibatch_share(8, 3, r) for r in range(3)] == [3, 3, 2]
    assert sum(minibatch_share(4096, 8, r) for r in range(8)) == 4096
