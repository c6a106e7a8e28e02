"""-m gpu: tools/secondary_shape_sweep.py -- seeded random shapes through the kernels of the secondary model types
(STDSEG_NO_DUR incl. more labels than lanes and more windows than wavefronts, STDSEG, the n-state frame model with phone
counts around the lane-group and wavefront boundaries), both precisions: gradient / numerator / Zx against the oracle
(1e-8 / 1e-9), lattice arcs bit for bit, Viterbi == the shortest path of the oracle's lattice."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_secondary_model_shape_sweep():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "secondary_shape_sweep.py"), "60", "3"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-4000:] + r.stderr[-2000:]
    assert r.stdout.count("\nok ") + r.stdout.startswith("ok ") == 60
