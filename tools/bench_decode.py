"""Decode throughput on BASELINE config 2 shape (GPU box): Viterbi best path for a resident batch
(scores in the reference's exact order + tropical recursion, labels bit-identical to the lattice
shortest path) and lattice-arc emission for single utterances.  Prints one JSON line."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "asr-craft_amd", "python"))
import numpy as np
import scrf_amd
from scrf_amd import synth

L, D, IN_W, T = 48, 25, 39, 300
U = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
F = 8 * IN_W + D
frames, labels, off = synth.make_batch(U, T, IN_W, L, D)
lam = synth.make_lambda(L * (F + 1 + L))
eng = scrf_amd.Engine(scrf_amd.make_config(L=L, D=D, F=F, precision=0, scratch_bytes=96 << 30)); eng.set_lambda(lam)
fl = [frames[int(off[u]):int(off[u + 1])] for u in range(U)]
b = eng.batch_from_frames(fl, None)
eng.viterbi_batch(b)                      # warm-up
eng.enable_timing(True)
t0 = time.perf_counter(); reps = 3
for _ in range(reps):
    labs, cost = eng.viterbi_batch(b)
dt = (time.perf_counter() - t0) / reps
tm = eng.last_timing()
eng.enable_timing(False)
t1 = time.perf_counter(); na = 0
for u in range(8):
    arcs, ns, fin = eng.lattice_arcs(b, u)
    na += len(arcs)
dta = (time.perf_counter() - t1) / 8
stats = eng.decode_stats()
print(json.dumps({"decode_path": "fused float weights + reference-order fix-ups" if eng.batch_is_fused(b) and os.environ.get("SCRF_FAST_DECODE", "1") != "0" else "exact scores",
                  "arc_weights_recomputed_per_batch": stats[0] // (reps + 1), "fallback_chunks": stats[1],
                  "metric": "utterances/sec SCRF Viterbi decode (TIMIT-shape)", "value": round(U / dt, 1), "unit": "utterances/s",
                  "utts": U, "ms_per_batch": round(1e3 * dt, 2),
                  "phase_ms": {k: round(v[0], 2) for k, v in tm.items() if v[0] > 0},
                  "lattice_arcs": {"arcs_per_utt": na // 8, "ms_per_utt_incl_copy_to_host": round(1e3 * dta, 2)}}))
