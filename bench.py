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


def host_topology():
    """(physical cores per socket, sockets, threads per core) from lscpu; None where it cannot be read"""
    import subprocess
    try:
        txt = subprocess.run(["lscpu"], capture_output=True, text=True, timeout=10).stdout
        kv = {k.strip(): v.strip() for k, v in (ln.split(":", 1) for ln in txt.splitlines() if ":" in ln)}
        return int(kv["Core(s) per socket"]), int(kv["Socket(s)"]), int(kv["Thread(s) per core"])
    except Exception:
        return None


def cpu_quota():
    """CPUs' worth of time the container's cgroup grants this process (cpu.max / cfs quota), or None when unlimited.
    The GPU boxes expose all 256 hardware threads but cap the cgroup at 16 CPUs: more runnable threads than that are
    throttled, not run (64 pinned workers measured 11.5x one worker there, with 439 of 870 periods throttled)."""
    try:
        q, p_ = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else float(q) / float(p_)
    except Exception:
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        p_ = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else q / p_
    except Exception:
        return None


def one_socket_cpus():
    """One logical CPU per physical core of the socket this process may use most of (`lscpu -p`), restricted to the
    CPUs this process is allowed on: the list the CPU baseline pins its workers to.  None if it cannot be read."""
    import subprocess
    try:
        allowed = os.sched_getaffinity(0)
        txt = subprocess.run(["lscpu", "-p=CPU,CORE,SOCKET"], capture_output=True, text=True, timeout=10).stdout
        per = {}
        for ln in txt.splitlines():
            if ln.startswith("#") or not ln.strip():
                continue
            cpu, core, sock = (int(x) if x else 0 for x in ln.split(",")[:3])
            if cpu in allowed:
                per.setdefault(sock, {}).setdefault(core, cpu)   # first hardware thread of each core
        if not per:
            return None
        best = max(per.values(), key=len)
        return sorted(best.values())
    except Exception:
        return None


def cpu_baseline(frames, labels, off, lam, cfg_kw=None, in_w=IN_W, frames2=None, in_w2=0, ctx2=0, utts_per_core=96):
    """The oracle (CPU restatement of the reference path, `port`) timed on this box's host cores
    over a bounded sample of the same workload: one worker thread per physical core of ONE socket,
    each PINNED to its core and keeping its arrays across utterances as the reference's threads keep their node
    vector (nodes/CRF_StateVector.cpp:37-67).  Reported beside the GPU number, never shipped."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orc
    try:
        allowed = len(os.sched_getaffinity(0))
    except AttributeError:
        allowed = os.cpu_count() or 1
    topo = host_topology()
    cpus = one_socket_cpus()
    quota = cpu_quota()
    cores = max(1, min(len(cpus) if cpus else allowed, topo[0] if topo else allowed, 128))
    if quota is not None:   # more workers than the cgroup's CPU quota would be throttled, not run
        cores = max(1, min(cores, int(quota)))
    orc.bench_set_cpus(cpus[:cores] if cpus else [])
    n = min(len(off) - 1, cores * utts_per_core)   # about 10-20 s of CPU work
    cfg = orc.config(**(cfg_kw or dict(L=L, D=D, F=F)))
    o = np.asarray(off[:n + 1])

    def run(nu, nthreads):
        oo = o[:nu + 1]
        if frames2 is None:
            return orc.bench_fb(cfg, lam, frames[:int(oo[-1])], labels[:int(oo[-1])], oo, in_w, nthreads)
        return orc.bench_fb2(cfg, lam, frames[:int(oo[-1])], frames2[:int(oo[-1]) + 2 * ctx2 * nu], in_w2, ctx2,
                             labels[:int(oo[-1])], oo, in_w, nthreads)
    rc, g, numer, zx, sec = run(n, cores)
    assert rc == 0
    ph = orc.bench_phases()
    tot = sum(ph.values()) or 1.0
    n1 = min(n, 2)
    rc1, _, _, _, sec1 = run(n1, 1)
    orc.bench_set_cpus([])
    single = n1 / sec1
    return {"value": round(n / sec, 3), "unit": "utterances/s", "cores": cores, "kind": "port",
            "pinned": bool(cpus), "cpu_list": (cpus[:cores] if cpus else None),
            "host": {"physical_cores_per_socket": topo[0] if topo else None, "sockets": topo[1] if topo else None,
                     "threads_per_core": topo[2] if topo else None, "cpus_allowed": allowed, "cgroup_cpu_quota": quota},
            # what a whole socket could reach at perfect scaling of the single-thread rate: an upper bound for the CPU
            # path on this host, extrapolated (the cgroup does not let this process measure it)
            "full_socket_upper_bound_utt_per_s": round(single * topo[0], 2) if topo else None,
            # the reference's own phase timers (gradbuilder :155-157, :481-488), microseconds per utterance and share
            "phase_us_per_utt": {k: round(v / n, 1) for k, v in ph.items()},
            "phase_share": {k: round(v / tot, 4) for k, v in ph.items()},
            "single_thread_utt_per_s": round(single, 3),
            "parallel_efficiency": round((n / sec) / (cores * single), 3),
            "sample": "%d utterances of the same batch, %d threads (one per physical core of one socket, pinned%s), %.1f s"
                      % (n, cores, "; the cgroup grants %.0f CPUs" % quota if quota is not None else "", sec)}, (g, numer, zx, n)


def kernels_fingerprint():
    """sha256 over the kernel sources: what a measured traffic figure belongs to"""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "asr-craft_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h", ".cpp")):
            h.update(f.encode()); h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def measured_traffic(section):
    """profiles/r04_traffic.json[section] if the file was measured on the kernel sources of this tree, else (None, why)"""
    tpath = os.path.join(ROOT, "profiles", "r04_traffic.json")
    if not os.path.exists(tpath):
        return None, "profiles/r04_traffic.json absent"
    with open(tpath) as fh:
        tj = json.load(fh)
    if tj.get("kernels_sha16") != kernels_fingerprint():
        return None, "profiles/r04_traffic.json was measured on other kernel sources (%s): not reported" % tj.get("kernels_sha16")
    return tj.get(section), None


def other_config(eng_mod, name, device_id, scratch_gib, with_cpu=True):
    """One forward-backward + gradient step at another BASELINE shape (after the timed region): BASELINE config 3
    (TIMIT demo: 48 labels, D = 10, 144-dim segment stream + +-6-frame context stream, `stdtrans`, 4 371 216 weights,
    demo/segmental-timit-demo.cfg.in:13-33) at 256 utterances of 304 frames, or the forward-backward half of config 5
    (200 labels, D = 40, 123-dim x 2000 frames) at 128 utterances.  Returns the entry of the line's `configs` array."""
    from scrf_amd import synth
    if name == "config3":
        Lc, Dc, W, T, U, ctx = 48, 10, 144, 304, 256, 6
    else:
        Lc, Dc, W, T, U, ctx = 200, 40, 123, 2000, 128, None
    rng = np.random.RandomState(4 if name == "config3" else 6)
    frames = [rng.random_sample((T, W)).astype(np.float32) for _ in range(U)]
    if name == "config3":   # rows L1-normalised like the MLP posteriors of the demo
        frames = [(f / f.sum(1, keepdims=True)).astype(np.float32) for f in frames]
    labels = [synth.group_labels(synth.frame_labels(rng, T, Lc, Dc), Dc, Lc) for _ in range(U)]
    Fs = 8 * W + Dc
    recipes = [eng_mod.StreamRecipe(W, 0, 0, 1)]
    streams2 = None
    kw = dict(L=Lc, D=Dc, F=Fs)
    Ft = 0
    if ctx:
        Ft = (2 * ctx + 1) * W
        kw = dict(L=Lc, D=Dc, F=Fs + Ft, sfe=Fs - 1, use_trans_ftrs=True, tfs=Fs)
        recipes.append(eng_mod.StreamRecipe(W, ctx, ctx, 0))
        streams2 = [[np.concatenate([np.repeat(f[:1], ctx, 0), f, np.repeat(f[-1:], ctx, 0)]) for f in frames]]
    eng = eng_mod.Engine(eng_mod.make_config(device_id=device_id, scratch_bytes=scratch_gib << 30, precision=3, **kw))
    lam = rng.normal(0, 0.01, eng.lambda_len)
    eng.set_lambda(lam)
    b = eng.batch_from_frames(frames, labels, recipes, streams2)
    eng.zero_grad(); eng.fb_batch(b, want_scalars=False); eng.synchronize()      # untimed first pass
    reps = 3
    t0 = time.perf_counter()
    for _ in range(reps):
        eng.zero_grad(); eng.fb_batch(b, want_scalars=False)
    eng.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / reps
    eng.enable_timing(True)
    eng.zero_grad(); eng.fb_batch(b, want_scalars=False); eng.synchronize()
    kt = sorted(eng.kernel_timing(), key=lambda x: -x[1])
    eng.enable_timing(False)
    nseg = n_segs(T, Dc)
    # F_alg = 4 N_seg L Fs' + (4T - 2) L^2 Ft' (feature counts without the bias)
    f_alg = 4.0 * nseg * Lc * Fs + (4.0 * T - 2) * Lc * Lc * Ft
    # flops the dense contraction kernels of the general path issue on the matrix pipe (2 per MAC)
    issued = {"k_scores_mfma(state)": 2.0 * nseg * Lc * (Fs + 1), "k_expf_mfma(state)": 2.0 * nseg * Lc * (Fs + 1),
              "k_scores_mfma(trans)": 2.0 * T * Lc * Lc * (Ft + 1), "k_expf_mfma(trans)": 2.0 * (T - 1) * Lc * Lc * (Ft + 1)}
    path = eng.batch_fused_mode(b)
    if path == 3:
        # hybrid path: the five sampled blocks (5 W of the 8 W + D columns) leave the dense contractions (per-frame
        # projections / per-frame sums); the count contraction also drops the one-hot duration and bias columns
        issued["k_scores_mfma(state)"] = 2.0 * nseg * Lc * (3 * W + Dc + 1)
        issued["k_expf_mfma(state)"] = 2.0 * nseg * Lc * (3 * W)
    dom_name, dom_ms, _ = kt[0]
    dom_flops = issued.get(dom_name)
    ent = {"name": name, "workload": "%d labels, D=%d, %d-dim x %d frames, %s, lambda_len %d, %d utterances per step"
                                     % (Lc, Dc, W, T, "stdtrans (+-%d context)" % ctx if ctx else "stdstate", eng.lambda_len, U),
           "ms_per_step": round(ms, 3), "utt_per_s": round(U / ms * 1e3, 1),
           "algorithmic_gflop_per_utt": round(f_alg / 1e9, 3),
           "frac_algorithmic": round(f_alg * U / (ms * 1e-3) / (PEAK["mfma_f64_tflops"] * 1e12), 4),
           "dominant_kernel": {"name": dom_name, "ms": round(dom_ms, 3),
                               "frac_executed": round(dom_flops * U / (dom_ms * 1e-3) / (PEAK["mfma_f64_tflops"] * 1e12), 4) if dom_flops else None},
           "path": {0: "general (materialised windows)", 1: "fused window synthesis", 2: "fused, linear window average",
                    3: "hybrid (materialised dense statistics; sampled blocks through per-frame projections and sums)"}.get(path, str(path)),
           "kernels_ms": {nm: round(m_, 3) for nm, m_, _ in kt[:8]}}
    # HBM bytes per step of this shape from the separate rocprofv3 --pmc passes (tools/collect_profiles.sh)
    tr, why = measured_traffic(name)
    ent["roofline"] = {"kernel": dom_name, "bound": "mfma" if dom_flops else "hbm",
                       "frac": ent["dominant_kernel"]["frac_executed"],
                       "traffic": round(tr["step_bytes"]) if tr else None,
                       # the trace names kernels by function (k_expf_mfma_ws + the single-role k_expf_mfma of remainder
                       # launches), the event table by launch site (k_expf_mfma(trans)): both forms of the dominant kernel
                       "traffic_dominant_kernel": round(sum(tr["kernels"].get(dom_name.split("(")[0] + sfx, {}).get("bytes", 0)
                                                            for sfx in ("", "_ws"))) if tr else None,
                       "traffic_note": why,
                       # B_alg = 4 T W_in + 3*8 (N_seg L + M_elems) + 2*8 (N_seg L + 2 T L) per utterance
                       "algorithmic_gb_per_step": round(U * (4.0 * T * (2 * W if ctx else W) + 24.0 * (nseg * Lc + (T * Lc * Lc if ctx else Lc * Lc))
                                                             + 16.0 * (nseg * Lc + 2 * T * Lc)) / 1e9, 2)}
    if with_cpu and name == "config3":
        # the north-star claim (>= 50x the reference's single-socket CPU rate on TIMIT-shape SCRF forward-backward) is
        # stated on THIS shape: the oracle's threaded path on a bounded sample, and the 1e-4 gate on the same sample
        fr = np.concatenate(frames); lb = np.concatenate(labels)
        off = np.concatenate([[0], np.cumsum([T] * U)]).astype(np.uint64)
        cb, (og, on, oz, n_cb) = cpu_baseline(fr, lb, off, lam, cfg_kw=kw, in_w=W, frames2=np.concatenate(streams2[0]), in_w2=W, ctx2=ctx,
                                              utts_per_core=8)
        ent["cpu_baseline"] = cb
        ent["speedup_vs_cpu_baseline"] = round(ent["utt_per_s"] / cb["value"], 1)
        if cb.get("full_socket_upper_bound_utt_per_s"):   # the north-star target (>= 50x one socket) against the most a socket could do
            ent["speedup_vs_full_socket_upper_bound"] = round(ent["utt_per_s"] / cb["full_socket_upper_bound_utt_per_s"], 1)
        eng.zero_grad()
        gb = eng.batch_from_frames(frames[:n_cb], labels[:n_cb], recipes, [streams2[0][:n_cb]])
        gn, gz = eng.fb_batch(gb)
        gg = eng.get_grad() / cb["cores"]
        gb.close()
        ent["parity_gate"] = {"utterances": int(n_cb), "tolerance": 1e-4,
                              "grad_rel": float(np.abs(gg - og).max() / np.abs(og).max()),
                              "numer_rel": float(np.abs(gn - on[:n_cb]).max() / max(1.0, np.abs(on[:n_cb]).max())),
                              "zx_rel": float(np.abs(gz - oz[:n_cb]).max() / np.abs(oz[:n_cb]).max())}
    b.close(); eng.close()
    return ent


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--utts", type=int, default=4096, help="utterances per rank per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-dist", action="store_true", help="initialise RCCL even for one rank (exercises the collective path)")
    ap.add_argument("--native-comm", action="store_true",
                    help="reduce the gradient through the ENGINE's own RCCL communicator (scrf_comm_init / scrf_allreduce_grad: what bin/CRFTrain "
                         "uses) instead of torch.distributed; the id travels by a torch broadcast")
    ap.add_argument("--precision", choices=["exact", "fast", "fast32", "fastlin"], default="fastlin",
                    help="exact: reference-order unfused fp64; fast: fp64 MFMA, window synthesis fused into the contractions; "
                         "fastlin (default): fast with the window average taken as the exact mean, which is linear in the frames and "
                         "leaves the dense contractions (fp64 throughout; 3e-8 on the gradient, contract 1e-4); fast32: f32 MFMA contractions (opt-in)")
    ap.add_argument("--scratch-gib", type=int, default=96, help="device scratch budget per chunk of utterances")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the config-3 / config-5 steps after the timed region")
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
                               precision={"exact": 0, "fast": 1, "fast32": 2, "fastlin": 3}[args.precision])
    eng = scrf_amd.Engine(cfg)
    lam = synth.make_lambda(eng.lambda_len)
    eng.set_lambda(lam)
    stream = torch.cuda.Stream()  # a real (non-null) HIP stream shared by the engine and torch/RCCL
    torch.cuda.set_stream(stream)
    eng.set_stream(stream.cuda_stream)
    from scrf_amd.dist import MinibatchReducer
    # gradient + {numer, zx, n_utts, active} in one device buffer: one RCCL all-reduce per step
    native = bool(args.native_comm and dist is not None)
    red = MinibatchReducer(eng.lambda_len, "cuda") if (dist is not None and not native) else None
    if native:
        # the engine's communicator: rank 0's id to every rank, then the collective init; the gradient stays in the engine
        uid = torch.zeros(128, dtype=torch.uint8, device="cuda")
        if rank == 0:
            uid.copy_(torch.frombuffer(bytearray(eng.comm_unique_id()), dtype=torch.uint8))
        dist.broadcast(uid, src=0)
        eng.comm_init(bytes(uid.cpu().numpy().tobytes()), rank, world)
    else:
        grad = red.grad if red is not None else torch.zeros(eng.lambda_len, dtype=torch.float64, device="cuda")
        eng.set_grad_buffer(grad.data_ptr())
    fl = [frames[int(off[u]):int(off[u + 1])] for u in range(U)]
    ll = [labels[int(off[u]):int(off[u + 1])] for u in range(U)]
    batch = eng.batch_from_frames(fl, ll)  # inputs resident in HBM from here on
    lr = 0.1 / U

    def step():
        eng.zero_grad()
        eng.fb_batch(batch, want_scalars=False)  # returns once the recursion's status is known; contractions in flight
        if native:
            eng.allreduce_grad(True)   # scrf_allreduce_grad: RCCL sum + / active on the engine's stream
        elif red is not None:
            # RCCL all-reduce (sum) of the weight gradient over xGMI, then / active ranks
            # (Minibatch_GradAccumulator.cpp:296-308); every rank is active in this bench
            red.reduce()
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
    gate_err = None
    if rank == 0:
        # ---- correctness gate, part 1: the timed steps left a finite model behind
        if not np.isfinite(eng.get_lambda()).all():
            gate_err = "lambda is not finite after the timed steps"
        # ---- roofline: three extra instrumented steps (HIP events on the engine's stream around every
        # kernel), kept outside the timed region because the per-kernel event waits would perturb `value`
        eng.set_lambda(lam)
        eng.enable_timing(True)
        acc, order = {}, []
        N_INSTR = 3
        phase = {}
        for _ in range(N_INSTR):
            eng.zero_grad()
            eng.fb_batch(batch, want_scalars=False)
            eng.synchronize()
            for name, ms_, nl_ in eng.kernel_timing():
                if name not in acc:
                    acc[name] = [0.0, 0]; order.append(name)
                acc[name][0] += ms_; acc[name][1] += nl_
            for k, (ms_, nl_) in eng.last_timing().items():
                phase[k] = phase.get(k, 0.0) + ms_
        eng.enable_timing(False)
        nseg = n_segs(T_FRAMES, D)
        nfr = T_FRAMES
        mfma_peak = PEAK["mfma_f32_tflops"] if args.precision == "fast32" else PEAK["mfma_f64_tflops"]
        # Work per utterance of every kernel of the step AS EXECUTED: flops the kernel issues
        # on the matrix pipe (2 per MAC) and the bytes its algorithm has to move (each array once per pass).
        # The fused kernels contract only the avg|max|min (+ one-hot, bias) column groups densely; the five
        # sampled blocks are re-associated into the per-frame contractions k_pframe / k_ztf.
        SL = 8.0 * nseg * L          # one pass over the [N_seg][L] fp64 array
        VL = 8.0 * nfr * L           # one [T][L] fp64 vector array
        # fastlin: a sixth per-frame group (the average) in k_pframe / k_ztf / k_post_z, two dense groups (max, min)
        # instead of three in the fused kernels; the wave-specialised count kernel (fast, fastlin) sums the one-hot
        # duration and bias columns instead of multiplying them
        la = args.precision == "fastlin"
        NG = 6 if la else 5
        GD = 2 if la else 3
        expf_cols = GD * IN_W if args.precision in ("fast", "fastlin") else 3 * IN_W + D + 1
        work = {
            "k_pframe": (2.0 * nfr * NG * L * IN_W, 4.0 * nfr * IN_W + NG * VL),
            "k_avg_prefix": (0.0, 2 * VL),
            "k_scores_fused": (2.0 * nseg * L * GD * IN_W, SL + 8.0 * nseg + NG * VL + 4.0 * nfr * IN_W),
            "k_windows": (0.0, 4.0 * nseg * F),
            "k_scores_exact(state)": (0.0, 4.0 * nseg * F + SL),      # fp64 VALU, unfused: priced by bytes only
            "k_scores_mfma(state)": (2.0 * nseg * L * F, 4.0 * nseg * F + SL),
            "k_true_scores": (0.0, 8.0 * nfr),
            "k_exp_rows": (0.0, 2 * SL),
            "k_dp_lin": (0.0, 2 * SL + 4 * VL),                         # ES read by both sweeps, 4 vectors written
            "k_dp_wave": (0.0, 3 * SL + 3 * VL),
            "k_post_z": (0.0, 2 * SL + NG * VL + 2 * VL),               # ES -> R in place, Z written, p and b read
            "k_post_lin": (0.0, 2 * SL + 2 * VL),
            "k_post_state": (0.0, 2 * SL + VL),
            "k_lin_z": (0.0, SL + 5 * VL),
            "k_mass_check": (0.0, 2 * VL),
            "k_expf_fused": (2.0 * nseg * L * expf_cols, SL + 4.0 * nfr * IN_W),
            "k_expf_mfma(state)": (2.0 * nseg * L * F, 4.0 * nseg * F + SL),
            "k_expf_gemm(state)": (0.0, 4.0 * nseg * F + SL),
            "k_ztf": (2.0 * nfr * NG * L * IN_W, NG * VL + 4.0 * nfr * IN_W),
            "reductions (k_reduce_slabs, k_atb, k_batch_sums)": (2.0 * nfr * L * L, 2 * VL),
        }
        kernels = []
        floors_mfma = floors_hbm = 0.0
        for name in order:
            ms_tot, nl = acc[name]
            ms_step = ms_tot / N_INSTR                          # this kernel's time per step
            flops_, bytes_ = work.get(name, (0.0, 0.0))
            t_mfma = flops_ * U / (mfma_peak * 1e12) * 1e3
            t_hbm = bytes_ * U / (PEAK["hbm_gbs"] * 1e9) * 1e3
            bound = "mfma" if t_mfma >= t_hbm else "hbm"
            floor = max(t_mfma, t_hbm)
            floors_mfma += t_mfma; floors_hbm += t_hbm
            ent = {"name": name, "ms": round(ms_step, 4), "launches_per_step": max(1, nl // N_INSTR), "bound": bound,
                   "tflop": round(flops_ * U / 1e12, 5), "gbyte": round(bytes_ * U / 1e9, 4),
                   "frac": round(floor / ms_step, 4) if ms_step > 0 and floor > 0 else None}
            kernels.append(ent)
        dom = max(kernels, key=lambda k: k["ms"])
        per_step = dom["launches_per_step"]
        avg_ms = dom["ms"] / per_step
        if dom["bound"] == "mfma":
            achieved = dom["tflop"] / per_step / (avg_ms / 1e3); peak = mfma_peak; unit = "TFLOP/s"
            per_launch = dom["tflop"] / per_step
        else:
            achieved = dom["gbyte"] / per_step / (avg_ms / 1e3); peak = PEAK["hbm_gbs"]; unit = "GB/s"
            per_launch = dom["gbyte"] / per_step
        # HBM bytes of the dominant kernel from separate rocprofv3 --pmc passes (profiles/README.md); the file names the
        # kernel sources it was measured on, and a figure from other sources is not reported
        traffic = None
        tsec, traffic_note = measured_traffic(args.precision)
        if tsec and tsec.get(dom["name"]):
            traffic = round(tsec[dom["name"]]["bytes_per_utt"] * U / per_step)
        frac = achieved / peak
        # SURVEY 8d's DENSE figure for the same kernel next to the executed one: the fused kernels contract only the
        # avg | max | min (+ one-hot, bias) columns and leave the five sampled blocks to k_pframe / k_ztf, so the dense
        # flops are not what they run (a fraction above 1 here is that re-association, not a faster pipe)
        dense = {"k_scores_fused": 2.0 * nseg * L * F, "k_expf_fused": 2.0 * nseg * L * F}.get(dom["name"])
        frac_alg = (dense * U / 1e12 / per_step / (avg_ms / 1e3) / peak) if (dense and dom["bound"] == "mfma") else frac
        step_ms = 1e3 * dt / args.steps
        # the whole step against its floors: matrix-pipe time of the flops the design executes, and SURVEY
        # 8d's algorithmic bytes per utterance (frames + 3 passes over S + 2 over the recursion arrays)
        b_alg = (4.0 * T_FRAMES * IN_W + 3 * 8.0 * (nseg * L + L * L) + 2 * 8.0 * (nseg * L + 2 * T_FRAMES * L)) * U
        hbm_floor = b_alg / (PEAK["hbm_gbs"] * 1e9) * 1e3
        roofline = {"kernel": dom["name"], "bound": dom["bound"], "achieved": round(achieved, 4), "peak": peak, "unit": unit,
                    "frac": round(frac, 5), "frac_algorithmic": round(frac_alg, 5), "traffic": traffic,
                    "traffic_note": traffic_note, "invalid": not (0.0 < frac <= 1.0),
                    "kernels_sha16": kernels_fingerprint(),
                    "work_per_launch": round(per_launch, 6), "work_is": "flops issued on the matrix pipe" if dom["bound"] == "mfma" else "algorithmic bytes",
                    "launches_per_step": per_step, "avg_launch_ms": round(avg_ms, 4),
                    "kernels": kernels,
                    "kernels_sum_ms": round(sum(k["ms"] for k in kernels), 3),
                    "phase_ms": {k: round(v / N_INSTR, 3) for k, v in phase.items() if not k.startswith("k_")},
                    "step": {"ms": round(step_ms, 3), "mfma_floor_ms": round(floors_mfma, 3),
                             "hbm_floor_ms": round(hbm_floor, 3), "algorithmic_gb": round(b_alg / 1e9, 2),
                             "frac_of_larger_floor": round(max(floors_mfma, hbm_floor) / step_ms, 4),
                             # SURVEY 8d's own bound for this config: the DENSE flops of scores + expected counts
                             # (4 N_seg L F per utterance, 0.466 GFLOP) on the fp64 matrix pipe = 169 k utterances/s per GPU
                             "survey_8d_dense_gflop_per_utt": round(4.0 * nseg * L * F / 1e9, 4),
                             "frac_of_survey_8d_mfma_bound": round(4.0 * nseg * L * F * U / (mfma_peak * 1e12) * 1e3 / step_ms, 4)}}
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
                       "parallelism": "dp%d" % world,
                       "rccl_ranks": dist.get_world_size() if dist is not None else 0,
                       "collective": ("engine RCCL communicator (scrf_allreduce_grad)" if native else "torch.distributed all_reduce (nccl = RCCL)") if dist is not None else None,
                       "step_sync": "scrf_fb_batch returns once the recursion's status is known (one event wait per step)"},
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
            cb, (og, on, oz, n_cb) = cpu_baseline(frames, labels, off, lam)
            out["cpu_baseline"] = cb
            out["speedup_vs_cpu_baseline"] = round(out["value"] / cb["value"], 1)
            if cb.get("full_socket_upper_bound_utt_per_s"):
                out["speedup_vs_full_socket_upper_bound"] = round(out["value"] / cb["full_socket_upper_bound_utt_per_s"], 1)
            # ---- correctness gate, part 2 (north_star: log-likelihood and gradients within 1e-4 relative): the
            # engine's gradient, numerators and log-partitions on the cpu_baseline sample against the oracle's.
            # The oracle's minibatch gradient is the sum over its worker streams / active streams
            # (CRF_Minibatch_GradAccumulator.cpp:296-308); every worker of the sample is active.
            eng.set_lambda(lam)
            eng.zero_grad()
            gb = eng.batch_from_frames(fl[:n_cb], ll[:n_cb])
            gn, gz = eng.fb_batch(gb)
            gg = eng.get_grad() / cb["cores"]
            gb.close()
            e_g = float(np.abs(gg - og).max() / np.abs(og).max())
            e_n = float(np.abs(gn - on[:n_cb]).max() / max(1.0, np.abs(on[:n_cb]).max()))
            e_z = float(np.abs(gz - oz[:n_cb]).max() / np.abs(oz[:n_cb]).max())
            out["parity_gate"] = {"utterances": int(n_cb), "grad_rel": e_g, "numer_rel": e_n, "zx_rel": e_z, "tolerance": 1e-4}
            if not (e_g <= 1e-4 and e_n <= 1e-4 and e_z <= 1e-4):
                gate_err = "engine vs oracle on the cpu_baseline sample: grad %.3g numer %.3g zx %.3g (tolerance 1e-4)" % (e_g, e_n, e_z)
        else:
            out["parity_gate"] = "skipped (--no-cpu-baseline): finiteness of lambda only"
    batch.close()
    eng.close()
    if rank == 0 and world == 1 and out is not None and not args.no_other_configs:
        # (at N = 1 only, like the CPU baseline: the other ranks would sit in the barrier below meanwhile)
        # the north-star training shape and the stress shape on the same clock (general path: materialised windows,
        # dense fp64-MFMA contractions); parity cases elsewhere, throughput entries here.  After the config-2 engine
        # has released its arena.
        out["configs"] = [other_config(scrf_amd, nm, local_rank, max(args.scratch_gib, 160), with_cpu=not args.no_cpu_baseline)
                          for nm in ("config3", "config5")]
        for ent in out["configs"]:
            pg = ent.get("parity_gate")
            if pg and not (pg["grad_rel"] <= 1e-4 and pg["numer_rel"] <= 1e-4 and pg["zx_rel"] <= 1e-4):
                gate_err = "%s: engine vs oracle grad %.3g numer %.3g zx %.3g (tolerance 1e-4)" % (ent["name"], pg["grad_rel"], pg["numer_rel"], pg["zx_rel"])
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    sys.stdout.flush()
    os.dup2(real_stdout, 1)
    if rank == 0:
        if gate_err:
            sys.stderr.write("bench.py: CORRECTNESS GATE FAILED: %s\n" % gate_err)
            sys.exit(3)
        print(json.dumps(out))
        sys.stdout.flush()


if __name__ == "__main__":
    main()
