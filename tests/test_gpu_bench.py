"""-m gpu: bench.py's line at a small size, through its three collective modes on one rank (no group,
torch.distributed's RCCL group, the engine's own RCCL communicator -- what bin/CRFTrain uses).  The
three runs do the same steps on the same data, so the line's contract fields and the correctness gate
must hold for each."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CONTRACT = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"]


def _run(extra, port):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
                        "--utts", "256", "--no-other-configs"] + extra, capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, "stdout must be exactly the JSON line:\n" + r.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.parametrize("mode", ["single", "torch-rccl", "engine-rccl"])
def test_bench_line_contract_in_every_collective_mode(mode):
    extra = {"single": [], "torch-rccl": ["--force-dist"], "engine-rccl": ["--force-dist", "--native-comm"]}[mode]
    # the CPU baseline (and with it the oracle gate) only on the first mode: it costs ~15 s of host time
    if mode != "single":
        extra = extra + ["--no-cpu-baseline"]
    line = _run(extra, 29531 + ["single", "torch-rccl", "engine-rccl"].index(mode))
    for k in CONTRACT:
        if k == "cpu_baseline" and mode != "single":
            continue
        assert k in line, k
    assert line["metric"].startswith("utterances/sec SCRF forward-backward") and line["unit"] == "utterances/s"
    assert line["n_gpus"] == 1 and line["steps"] == 2 and line["warmup"] == 1
    assert line["dtype"] == "f64" and line["scaling"] == "weak" and line["higher_is_better"] is True
    assert line["value"] > 0 and abs(line["value"] - 256 / (line["ms_per_step"] * 1e-3)) <= 0.01 * line["value"]
    rf = line["roofline"]
    assert rf["bound"] in ("hbm", "mfma") and 0 < rf["frac"] <= 1.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    assert line["config"]["rccl_ranks"] == (0 if mode == "single" else 1)
    if mode == "single":
        cb = line["cpu_baseline"]
        assert cb["kind"] == "port" and cb["value"] > 0 and cb["cores"] >= 1
        g = line["parity_gate"]
        assert g["grad_rel"] <= g["tolerance"] and g["zx_rel"] <= 1e-8
    if mode == "engine-rccl":
        assert line["config"]["collective"].startswith("engine")
