#!/usr/bin/env python3
"""bench.py -- utterances/sec of segmental-CRF forward-backward (+ gradient reduce + SGD step)
on synthetic TIMIT-shape feature streams, BASELINE.json config 2:
  48 labels, max segment length 25, 39-dim x 300-frame utterances, `stdstate` feature map.

One "step" = one SGD minibatch on every rank: forward-backward + expected-count gradient over
the rank's utterances (already resident in HBM), all-reduce (sum) of the weight gradient over
RCCL, divide by the active ranks, SGD update.  Weak scaling: every rank processes --utts
utterances per step.  Prints ONE JSON line on rank 0.

  python bench.py --gpus 1 --steps 5 --warmup 2
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "asr-craft_amd", "python"))

import numpy as np  # noqa: E402

L, D, IN_W, T_FRAMES = 48, 25, 39, 300
F = 8 * IN_W + D  # 337 segment features (io/CRF_InFtrStream_SeqMultiWindow.cpp:77-78)
PEAK = {"mfma_f64_tflops": 78.6, "mfma_f32_tflops": 157.3, "hbm_gbs": 8000.0}  # MI355X_MICROARCH.md


def n_segs(T, Dm):
    return T * (T + 1) // 2 if T < Dm else Dm * (Dm + 1) // 2 + (T - Dm) * Dm


def cpu_baseline(frames, labels, off, lam):
    """The oracle (CPU restatement of the reference path, `port`) timed on this box's host cores
    over a bounded sample of the same workload.  Reported beside the GPU number, never shipped."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orc
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 64))
    n = min(len(off) - 1, cores * 24)
    cfg = orc.config(L=L, D=D, F=F)
    o = np.asarray(off[:n + 1])
    rc, g, numer, zx, sec = orc.bench_fb(cfg, lam, frames[:int(o[-1])], labels[:int(o[-1])], o, IN_W, cores)
    assert rc == 0
    rc1, _, _, _, sec1 = orc.bench_fb(cfg, lam, frames[:int(o[2])], labels[:int(o[2])], o[:3], IN_W, 1)
    return {"value": round(n / sec, 3), "unit": "utterances/s", "cores": cores, "kind": "port",
            "sample": "%d utterances of the same batch, %d threads, %.1f s; single thread %.2f utt/s"
                      % (n, cores, sec, 2 / sec1)}, (g, numer, zx, n)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--utts", type=int, default=4096, help="utterances per rank per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-dist", action="store_true", help="initialise RCCL even for one rank (exercises the collective path)")
    ap.add_argument("--precision", choices=["exact", "fast", "fast32"], default="fast",
                    help="exact: reference-order unfused fp64; fast: fp64 MFMA, window synthesis fused into the contractions; fast32: f32 MFMA contractions (opt-in)")
    ap.add_argument("--scratch-gib", type=int, default=96, help="device scratch budget per chunk of utterances")
    args = ap.parse_args()

    # the JSON line must be the only thing on stdout: until it is printed, file descriptor 1 points
    # at stderr, so that library banners (RCCL prints one under NCCL_DEBUG=VERSION) do not mix in
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    import torch
    import scrf_amd
    from scrf_amd import synth

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the engine has no CPU path")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    U = args.utts
    # synthetic data of the config-2 shape; every rank owns a different contiguous utterance range
    frames, labels, off = synth.make_batch(U, T_FRAMES, IN_W, L, D, seed=1234 + 100003 * rank)
    cfg = scrf_amd.make_config(L=L, D=D, F=F, device_id=local_rank, scratch_bytes=args.scratch_gib << 30,
                               precision={"exact": 0, "fast": 1, "fast32": 2}[args.precision])
    eng = scrf_amd.Engine(cfg)
    lam = synth.make_lambda(eng.lambda_len)
    eng.set_lambda(lam)
    stream = torch.cuda.Stream()  # a real (non-null) HIP stream shared by the engine and torch/RCCL
    torch.cuda.set_stream(stream)
    eng.set_stream(stream.cuda_stream)
    grad = torch.zeros(eng.lambda_len, dtype=torch.float64, device="cuda")
    sums = torch.zeros(4, dtype=torch.float64, device="cuda")
    eng.set_grad_buffer(grad.data_ptr())
    fl = [frames[int(off[u]):int(off[u + 1])] for u in range(U)]
    ll = [labels[int(off[u]):int(off[u + 1])] for u in range(U)]
    batch = eng.batch_from_frames(fl, ll)  # inputs resident in HBM from here on
    lr = 0.1 / U

    from scrf_amd.dist import reduce_minibatch

    def step():
        eng.zero_grad()
        eng.fb_batch(batch, want_scalars=False)  # async on the shared stream
        if dist is not None:
            # RCCL all-reduce (sum) of the weight gradient over xGMI, then / active ranks
            # (Minibatch_GradAccumulator.cpp:296-308); every rank is active in this bench
            reduce_minibatch(grad, sums[:3], True)
        eng.sgd_step(lr, False)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    out = None
    if rank == 0:
        # ---- roofline of the dominant kernel: three extra instrumented steps (HIP events on the
        # engine's stream around every phase and around the three big kernels), kept outside the
        # timed region because the per-phase event waits would perturb `value`
        eng.set_lambda(lam)
        eng.enable_timing(True)
        acc = {}
        N_INSTR = 3
        for _ in range(N_INSTR):
            eng.zero_grad()
            eng.fb_batch(batch, want_scalars=False)
            eng.synchronize()
            for k, (ms_, nl_) in eng.last_timing().items():
                a_ = acc.setdefault(k, [0.0, 0])
                a_[0] += ms_; a_[1] += nl_
        tm = {k: (v[0], v[1]) for k, v in acc.items()}
        eng.enable_timing(False)
        nseg = n_segs(T_FRAMES, D)
        # algorithmic work per utterance: dense flops of the two state contractions
        # (2 per MAC, bias excluded); bytes the recursion has to move (S read by the forward and
        # by the backward sweep, alpha / alpha-plus-trans / beta / sum-over-durations written)
        flops_gemm = 2.0 * nseg * L * F
        bytes_dp = 8.0 * (2 * nseg * L + 4 * T_FRAMES * L)
        mfma_peak = PEAK["mfma_f32_tflops"] if args.precision == "fast32" else PEAK["mfma_f64_tflops"]
        kern = {
            "k_scores": ("k_scores_fused" if args.precision != "exact" else "k_scores_exact", "mfma",
                         flops_gemm / 1e12, "TFLOP/s", mfma_peak),
            "k_expf": ("k_expf_fused" if args.precision != "exact" else "k_expf_gemm", "mfma",
                       flops_gemm / 1e12, "TFLOP/s", mfma_peak),
            "k_dp": ("k_dp_lin", "hbm", bytes_dp / 1e9, "GB/s", PEAK["hbm_gbs"]),
        }
        dom = max(kern, key=lambda k: tm[k][0])
        kname, bound, work_per_utt, unit, peak = kern[dom]
        ms, nl = tm[dom]
        nl = max(1, int(nl))
        avg_ms = ms / nl                       # average duration of one launch of that kernel
        per_step = max(1, nl // N_INSTR)       # launches per step (1 unless the batch had to be chunked)
        work = work_per_utt * U / per_step     # algorithmic work of one launch
        achieved = work / (avg_ms / 1e3)
        # what the fused kernels actually execute on the MFMA: the sampled-frame blocks are
        # re-associated into per-frame projections, so only 3W (scores) / 3W+D+1 (counts) of the
        # 8W+D columns go through the dense product (DESIGN.md "roofline accounting")
        executed = None
        if bound == "mfma" and args.precision != "exact":
            cols = 3 * IN_W if dom == "k_scores" else 3 * IN_W + D + 1
            executed = 2.0 * nseg * L * cols * U / per_step / 1e12
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "r01_traffic.json")
        if os.path.exists(tpath):   # HBM bytes per utterance from separate rocprofv3 --pmc passes (profiles/README.md)
            with open(tpath) as fh:
                tj = json.load(fh)
            ent = tj.get(args.precision, {}).get(kname)
            if ent:
                traffic = round(ent["bytes_per_utt"] * U / per_step)
        roofline = {"kernel": kname, "bound": bound, "achieved": round(achieved, 4), "peak": peak, "unit": unit,
                    "frac": round(achieved / peak, 5), "traffic": traffic,
                    "algorithmic_per_launch": round(work, 6), "launches_per_step": per_step,
                    "avg_launch_ms": round(avg_ms, 4),
                    "kernel_ms": {k: round(tm[k][0] / max(1, tm[k][1]), 3) for k in kern},
                    "phase_ms": {k: round(v[0] / N_INSTR, 3) for k, v in tm.items() if not k.startswith("k_")}}
        if executed is not None:
            roofline["executed_per_launch"] = round(executed, 6)
            roofline["frac_executed"] = round(executed / (avg_ms / 1e3) / peak, 5)
        # the whole step against SURVEY 8d's per-utterance totals (F_alg = two dense contractions,
        # B_alg = frames + 3 passes over S + 2 over the recursion arrays): rank-local, timed region
        step_s = dt / args.steps
        f_alg = 2.0 * flops_gemm * U
        b_alg = (4.0 * T_FRAMES * IN_W + 3 * 8.0 * (nseg * L + L * L) + 2 * 8.0 * (nseg * L + 2 * T_FRAMES * L)) * U
        roofline["step"] = {"algorithmic_tflop": round(f_alg / 1e12, 4), "tflops": round(f_alg / 1e12 / step_s, 2),
                            "frac_mfma": round(f_alg / 1e12 / step_s / mfma_peak, 4),
                            "algorithmic_gb": round(b_alg / 1e9, 2), "frac_hbm": round(b_alg / 1e9 / step_s / PEAK["hbm_gbs"], 4)}
        out = {
            "metric": "utterances/sec SCRF forward-backward (TIMIT-shape)",
            "value": round(U * world * args.steps / dt, 2),
            "unit": "utterances/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * dt / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "BASELINE config 2: segmental CRF forward-backward, 48 labels, max-seg-len 25, "
                                   "39-dim x 300-frame utterances, stdstate map (lambda_len 18528)",
                       "utts_per_rank_per_step": U, "global_minibatch": U * world, "precision": args.precision,
                       "parallelism": "dp%d" % world},
            "roofline": roofline,
        }
        # decode half of the metric: best-path labels of the same resident batch, bit-identical
        # to the reference-order evaluation (fp64-MFMA arc weights + reference-order recomputation of the
        # entries a rounding-error bound cannot clear); one untimed call first
        eng.viterbi_batch(batch)
        eng.synchronize()
        t1 = time.perf_counter()
        eng.viterbi_batch(batch)
        eng.synchronize()
        dtv = time.perf_counter() - t1
        out["decode"] = {"metric": "utterances/sec SCRF Viterbi decode (TIMIT-shape), labels copied to host",
                         "value": round(U / dtv, 1), "unit": "utterances/s", "ms_per_batch": round(1e3 * dtv, 2),
                         "dtype": "f64 scores (MFMA + reference-order fix-ups), f32 tropical recursion",
                         "arc_weights_recomputed": eng.decode_stats()[0] // 2, "fallback_chunks": eng.decode_stats()[1]}
        if not args.no_cpu_baseline:
            cb, _ = cpu_baseline(frames, labels, off, lam)
            out["cpu_baseline"] = cb
            out["speedup_vs_cpu_baseline"] = round(out["value"] / cb["value"], 1)
    batch.close()
    eng.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    sys.stdout.flush()
    os.dup2(real_stdout, 1)
    if rank == 0:
        print(json.dumps(out))
        sys.stdout.flush()


if __name__ == "__main__":
    main()
