"""Per-kernel HIP-event times of one config-2 training step (the engine's own instrumentation, scrf_kernel_timing),
without the timed region, the CPU baseline and the correctness gate of bench.py: for A/B experiments on kernels whose
results may be deliberately wrong (ablations).  tools/ktimes.py [--utts 4096] [--reps 3]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "asr-craft_amd", "python"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--utts", type=int, default=4096)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--precision", default="fast")
    args = ap.parse_args()
    import torch
    import scrf_amd
    from scrf_amd import synth
    L, D, IN_W, T = 48, 25, 39, 300
    frames, labels, off = synth.make_batch(args.utts, T, IN_W, L, D, seed=1234)
    cfg = scrf_amd.make_config(L=L, D=D, F=8 * IN_W + D, device_id=0, scratch_bytes=96 << 30,
                               precision={"exact": 0, "fast": 1, "fast32": 2, "fastlin": 3}[args.precision])
    eng = scrf_amd.Engine(cfg)
    eng.set_lambda(synth.make_lambda(eng.lambda_len))
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    eng.set_stream(stream.cuda_stream)
    U = args.utts
    batch = eng.batch_from_frames([frames[int(off[u]):int(off[u + 1])] for u in range(U)],
                                  [labels[int(off[u]):int(off[u + 1])] for u in range(U)])
    for _ in range(2):
        eng.zero_grad()
        try:
            eng.fb_batch(batch, want_scalars=False)
        except Exception as e:  # ablations may trip the status checks
            print("note:", str(e)[:100])
        eng.synchronize()
    eng.enable_timing(True)
    acc, order = {}, []
    for _ in range(args.reps):
        eng.zero_grad()
        try:
            eng.fb_batch(batch, want_scalars=False)
        except Exception:
            pass
        eng.synchronize()
        for name, ms_, nl_ in eng.kernel_timing():
            if name not in acc:
                acc[name] = 0.0
                order.append(name)
            acc[name] += ms_
    tot = 0.0
    for name in order:
        print("  %-50s %8.3f ms" % (name, acc[name] / args.reps))
        tot += acc[name] / args.reps
    print("  %-50s %8.3f ms" % ("sum", tot))


if __name__ == "__main__":
    main()
