"""profiles/<tag>_traffic.json from the per-kernel FETCH_SIZE / WRITE_SIZE passes (tools/pmc_kernels.sh output):
HBM bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) KB -- gfx950 tallies 128-byte read requests at 64 bytes
(MI355X_MICROARCH.md, "HBM").  usage: python tools/make_traffic_json.py <round tag, e.g. r01> [utterances per launch]"""
import json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
U = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
KEEP = ("k_scores_fused", "k_dp_lin", "k_post_z", "k_expf_fused", "k_pframe", "k_ztf", "k_mass_check")


def parse(path):
    out, cur = {}, None
    for line in open(path):
        m = re.match(r"^(?:void )?(k_\w+)(<[^>]*>)?", line)
        if m and "launches" in line:
            cur = m.group(1)
            # the decode variant k_scores_fused<D, 0, 1> also runs once in bench.py: keep the training kernel
            if cur == "k_expf_fused_ws":   # the wave-specialised build of the same kernel
                cur = "k_expf_fused"
            if cur == "k_scores_fused" and m.group(2) and m.group(2).rstrip(">").split(",")[-1].strip() == "1":
                cur = "k_scores_fused_decode"
        elif cur and line.strip():
            name, val = line.split()
            out.setdefault(cur, {})[name] = float(val)
    return out


res = {}
for prec in ("fast", "fast32"):
    pf = os.path.join(ROOT, "profiles", "%s_%s_pmc_fetch.txt" % (tag, prec))
    pw = os.path.join(ROOT, "profiles", "%s_%s_pmc_write.txt" % (tag, prec))
    if not (os.path.exists(pf) and os.path.exists(pw)):
        continue
    f = parse(pf)
    w = parse(pw)
    res[prec] = {}
    for k in KEEP:
        if k not in f or k not in w:
            continue
        fr, wr = f[k]["FETCH_SIZE"], w[k]["WRITE_SIZE"]
        b = int((2 * fr + wr) * 1024)
        res[prec][k] = {"fetch_size_kb_raw": fr, "write_size_kb": wr, "bytes_per_launch": b, "bytes_per_utt": b / U}
# the figures belong to the kernel sources they were measured on: bench.py reports them only while that fingerprint
# (its kernels_fingerprint(): sha256 over asr-craft_amd/csrc/*.{hip,h,cpp}) is the tree's
sys.path.insert(0, ROOT)
import bench
res["kernels_sha16"] = bench.kernels_fingerprint()
try:
    import subprocess
    res["commit"] = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or None
except Exception:
    res["commit"] = None
json.dump(res, open(os.path.join(ROOT, "profiles", "%s_traffic.json" % tag), "w"), indent=1)
print(json.dumps({p: {k: round(v["bytes_per_launch"] / 1e9, 2) for k, v in d.items()} for p, d in res.items() if isinstance(d, dict)}))
