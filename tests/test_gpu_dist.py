"""-m gpu: the data-parallel layer with ENGINE gradients.  Two processes (ranks) each own an engine and
the contiguous utterance view of their rank (io/CRF_FeatureStreamManager.cpp:425-464), run their share of
every minibatch through scrf_fb_batch, and join with an all-reduce (sum) / active ranks
(CRF_Minibatch_GradAccumulator.cpp:277-312).  The one-GPU test box cannot host two RCCL ranks (RCCL wants
one GPU per rank), so the collective here is gloo on the host copies of the device gradients; the RCCL
path itself is covered at one rank (test_native_rccl_single_rank, the CRFTrain communicator test) and by the
driver's multi-GPU bench.  Reference result: ONE engine walking the same streams in stream order."""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _problem():
    sys.path.insert(0, os.path.join(ROOT, "asr-craft_amd", "python"))
    import scrf_amd
    from scrf_amd import synth
    L, D, W, U = 6, 4, 3, 9
    frames, labels, off = synth.make_batch(U, 14, W, L, D, seed=77, t_jitter=4)
    F = 8 * W + D
    cfg = scrf_amd.make_config(L=L, D=D, F=F, precision=1)
    lam = synth.make_lambda(L * (F + 1 + L), scale=0.2)
    fl = [frames[int(off[u]):int(off[u + 1])] for u in range(U)]
    ll = [labels[int(off[u]):int(off[u + 1])] for u in range(U)]
    return scrf_amd, cfg, lam, fl, ll, U


def _worker(rank, world, port, minibatch, out_dir):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    scrf_amd, cfg, lam, fl, ll, U = _problem()
    from scrf_amd.dist import MinibatchReducer, RankCursor
    eng = scrf_amd.Engine(cfg); eng.set_lambda(lam)
    cur = RankCursor(U, world, rank)
    red = MinibatchReducer(eng.lambda_len, "cpu")
    steps = []
    while True:
        idx = list(cur.next_step(minibatch))
        eng.zero_grad()
        if idx:
            b = eng.batch_from_frames([fl[u] for u in idx], [ll[u] for u in idx])
            eng.fb_batch(b, want_scalars=False)
            b.close()
        red.grad.copy_(torch.from_numpy(eng.get_grad()))
        red.tail[:3] = torch.from_numpy(eng.batch_sums())
        red.set_active(bool(idx))
        if int(red.reduce().item()) == 0:
            break
        steps.append(np.concatenate([red.grad.numpy(), red.tail.numpy()]))
    if rank == 0:
        np.save(os.path.join(out_dir, "steps.npy"), np.array(steps))
    eng.close()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("minibatch", [3, 4])
def test_two_ranks_of_engine_gradients_equal_one_engine_in_stream_order(tmp_path, minibatch):
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    world = 2
    mp.spawn(_worker, args=(world, port, minibatch, str(tmp_path)), nprocs=world, join=True)
    got = np.load(os.path.join(str(tmp_path), "steps.npy"))
    scrf_amd, cfg, lam, fl, ll, U = _problem()
    from scrf_amd.dist import RankCursor
    eng = scrf_amd.Engine(cfg); eng.set_lambda(lam)
    curs = [RankCursor(U, world, r) for r in range(world)]
    k = 0
    while any(c.active for c in curs):
        tot = np.zeros(eng.lambda_len); sums = np.zeros(3); active = 0
        for c in curs:
            idx = list(c.next_step(minibatch))
            if not idx:
                continue
            active += 1
            eng.zero_grad()
            b = eng.batch_from_frames([fl[u] for u in idx], [ll[u] for u in idx])
            eng.fb_batch(b, want_scalars=False)
            tot += eng.get_grad(); sums += eng.batch_sums()
            b.close()
        n = eng.lambda_len
        np.testing.assert_allclose(got[k, :n], tot / active, rtol=1e-12, atol=1e-13 * np.abs(tot).max())
        np.testing.assert_allclose(got[k, n:n + 3], sums, rtol=1e-12)
        assert got[k, n + 3] == active
        k += 1
    assert k == got.shape[0]
    eng.close()
