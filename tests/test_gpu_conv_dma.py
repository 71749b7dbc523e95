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


def test_sync_batchnorm_two_ranks_match_one_process():
    """set_sync_batchnorm(True): two ranks (gloo, sharing this GPU) with half a batch each reproduce the single-process
    whole-batch logits, loss, running statistics and reduced gradients (tools/check_syncbn.py)."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, DML_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), os.path.join(H.ROOT, "tools", "check_syncbn.py")], env=env,
                       capture_output=True, text=True, cwd=H.ROOT, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "rank 0:" in r.stdout and "rank 1:" in r.stdout
