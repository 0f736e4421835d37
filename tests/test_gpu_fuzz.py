"""A bounded run of the random-shape screen (tools/fuzz_shapes.py: layer counts 1-4, kernel sizes 1-7, ragged grids, 1-62 input
channels, B = 1-5, T = 1-4, both storage types, merged-grid launches on / off, both tile heights) against the CPU oracle:
the module / autograd path, the fused trainer path, the single cell with a given state, and the device preproc of a
resident record (as a batch tensor and straight into the model's input slab).  Seeds are fixed; the tool itself takes any seed.  (It found the
one-input-channel fold bug of round 3.)"""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("extra", [["--n", "30", "--seed", "11"], ["--n", "16", "--seed", "12", "--trainer"], ["--n", "24", "--seed", "13", "--cell"], ["--n", "24", "--seed", "14", "--dataset"]])
def test_random_shapes_against_the_oracle(extra):
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_shapes.py")] + extra, capture_output=True, text=True, timeout=600)
    tail = "\n".join((r.stdout + r.stderr).splitlines()[-6:])
    assert r.returncode == 0, tail
    assert "shapes ok" in r.stdout, tail
    print(r.stdout.splitlines()[-1])
