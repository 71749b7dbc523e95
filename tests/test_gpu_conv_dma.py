"""GPU: the LDS-DMA conv kernel (buffer_load ... lds ring) is only selected automatically for very large grids;
force it with DML_CONV_DMA=1 in a child process and run the conv parity tests through it."""
import os
import subprocess
import sys

import pytest

import helpers as H

pytestmark = pytest.mark.gpu


def test_conv_ops_through_the_dma_kernel():
    env = dict(os.environ, DML_CONV_DMA="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(H.ROOT, "tests", "test_gpu_ops.py"), "-m", "gpu",
                        "-q", "-x", "-k", "conv", "-p", "no:cacheprovider"], env=env, capture_output=True, text=True,
                       cwd=H.ROOT, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "passed" in r.stdout
