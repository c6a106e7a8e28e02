"""profiles/<tag>_traffic.json from the FETCH_SIZE / WRITE_SIZE passes of tools/collect_profiles.sh (tools/pmc_pass.sh
output): HBM bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) KB -- gfx950 tallies 128-byte read requests at 64 bytes
(MI355X_MICROARCH.md, "HBM").
  python tools/make_traffic_json.py <tag> [precision] [utterances per launch]
Sections: <precision> (bench.py's config-2 step: per kernel bytes per launch / per utterance), config3, config5 (one
forward-backward step of tools/prof_shape.py cfg3x256 / cfg5x128: bytes per step and per kernel).
Every pass file starts with `# kernels_sha16 <fingerprint>` written AT MEASUREMENT TIME; passes whose fingerprints are
missing or differ are refused, and the json carries that fingerprint (not the tree's at the time this script runs):
bench.py reports the figures only while it equals the tree's."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
prec = sys.argv[2] if len(sys.argv) > 2 else "fastlin"
U = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
P = os.path.join(ROOT, "profiles")


def parse(path):
    """-> (fingerprint, {kernel: {"launches": n, counter: mean per launch}})"""
    sha, out, cur = None, {}, None
    for line in open(path):
        m = re.match(r"^# kernels_sha16 (\w+)", line)
        if m:
            sha = m.group(1)
            continue
        m = re.match(r"^(?:void )?(k_\w+)(<[^>]*>)?.* launches (\d+)", line)
        if m:
            cur = m.group(1)
            targs = (m.group(2) or "").strip("<>").replace(" ", "").split(",")
            if cur == "k_expf_fused_ws":   # the wave-specialised build of the same kernel
                cur = "k_expf_fused"
            # the decode form k_scores_fused<D, 0, 1, 0> also runs in bench.py: keep the training kernel apart
            if cur == "k_scores_fused" and len(targs) >= 3 and targs[2] == "1":
                cur = "k_scores_fused_decode"
            out.setdefault(cur, {"launches": 0})
            out[cur]["launches"] += int(m.group(3))
            out[cur]["_n"] = int(m.group(3))
        elif cur and line.strip() and not line.startswith("#"):
            name, val = line.split()
            # mean per launch over the (possibly several) template instances folded into one name
            out[cur][name] = out[cur].get(name, 0.0) + float(val) * out[cur]["_n"]
    for k in out.values():
        n = k.pop("_n", None)
        for c in list(k):
            if c != "launches":
                k[c] /= k["launches"]
    return sha, out


def section(stem):
    pf, pw = os.path.join(P, "%s_%s_pmc_fetch.txt" % (tag, stem)), os.path.join(P, "%s_%s_pmc_write.txt" % (tag, stem))
    if not (os.path.exists(pf) and os.path.exists(pw)):
        return None, None
    sf, f = parse(pf)
    sw, w = parse(pw)
    if not sf or not sw or sf != sw:
        sys.exit("make_traffic_json: %s / %s carry fingerprints %s / %s: measure both passes on one tree" % (pf, pw, sf, sw))
    res = {}
    for k in f:
        if k in w and "FETCH_SIZE" in f[k] and "WRITE_SIZE" in w[k]:
            fr, wr = f[k]["FETCH_SIZE"], w[k]["WRITE_SIZE"]
            res[k] = {"fetch_size_kb_raw": fr, "write_size_kb": wr, "launches": f[k]["launches"], "bytes_per_launch": int((2 * fr + wr) * 1024)}
    return sf, res


out = {}
shas = set()
sha, k2 = section(prec)
if k2:
    shas.add(sha)
    out[prec] = {k: dict(v, bytes_per_utt=v["bytes_per_launch"] / U) for k, v in k2.items()}
for stem, name, steps in (("cfg3", "config3", 2), ("cfg5", "config5", 2)):   # prof_shape.py <shape> 1 runs 1 + 1 steps
    sha, ks = section(stem)
    if not ks:
        continue
    shas.add(sha)
    kern = {}
    for k, v in ks.items():
        per_step = v["bytes_per_launch"] * v["launches"] / steps
        kern[k] = {"bytes": per_step, "launches_per_step": v["launches"] / steps}
    out[name] = {"step_bytes": sum(x["bytes"] for x in kern.values()), "kernels": kern}
if len(shas) != 1:
    sys.exit("make_traffic_json: the passes were measured on %d different kernel source trees (%s)" % (len(shas), sorted(shas)))
out["kernels_sha16"] = shas.pop()
json.dump(out, open(os.path.join(P, "%s_traffic.json" % tag), "w"), indent=1)
print(json.dumps({p: ({k: round(v["bytes_per_launch"] / 1e9, 2) for k, v in d.items()} if p == prec else round(d["step_bytes"] / 1e9, 2))
                  for p, d in out.items() if isinstance(d, dict)}))
