"""Step time of the BASELINE config-3 and config-5 shapes (bench.py's `configs` entries) alone, for A/B runs of the
general-path kernels:  python tools/time_cfg.py [config3|config5 ...]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "asr-craft_amd", "python"))
import scrf_amd
import bench
for nm in (sys.argv[1:] or ["config3", "config5"]):
    e = bench.other_config(scrf_amd, nm, 0, 160)
    print(nm, e["ms_per_step"], "ms", e["utt_per_s"], "utt/s", "frac_alg", e["frac_algorithmic"], json.dumps(e["kernels_ms"]))
