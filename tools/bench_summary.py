#!/usr/bin/env python3
"""Readable summary of a bench.py JSON line: tools/bench_summary.py gpurun_out/bench.json"""
import json
import sys

d = json.load(open(sys.argv[1]))
print("value %.1f utt/s  %.3f ms/step  precision %s  n_gpus %d" % (d["value"], d["ms_per_step"], d["config"]["precision"], d["n_gpus"]))
r = d["roofline"]
print("dominant %s frac %.3f frac_alg %.3f traffic %s" % (r["kernel"], r["frac"], r["frac_algorithmic"], r.get("traffic")))
print("step", r["step"])
for k in r["kernels"]:
    print("   %-52s %8.3f ms  %-4s frac %s" % (k["name"], k["ms"], k["bound"], k["frac"]))
if "decode" in d:
    print("decode %.0f utt/s (%.2f ms)" % (d["decode"]["value"], d["decode"]["ms_per_batch"]))
skip = ("host", "cpu_list")
if "cpu_baseline" in d:
    print("cpu", {k: v for k, v in d["cpu_baseline"].items() if k not in skip}, "speedup", d.get("speedup_vs_cpu_baseline"))
print("gate", d.get("parity_gate"))
for c in d.get("configs", []):
    print("%s: %.3f ms  %.1f utt/s  frac_alg %.3f  dominant %s  speedup_vs_cpu %s" % (
        c["name"], c["ms_per_step"], c["utt_per_s"], c["frac_algorithmic"], c["dominant_kernel"], c.get("speedup_vs_cpu_baseline")))
    if "cpu_baseline" in c:
        print("    cpu", {k: v for k, v in c["cpu_baseline"].items() if k not in skip})
    if "parity_gate" in c:
        print("    gate", c["parity_gate"])
    print("    roofline", c.get("roofline"))
    print("    kernels", c["kernels_ms"])
