"""GPU: the product's sharded batch entry points under a real RCCL process group.  A 1-GPU box can only form a
world of one rank, which still exercises the whole wiring: sharded_map -> HIP kernels through the C ABI ->
ragged all-gather on the device over backend "nccl" (= RCCL).  The world-size-2 logic is covered on CPU with gloo
(tests/test_distributed_cpu.py).  Runs in a fresh interpreter (tests/rccl_worker.py): torch has to be imported before
liblkhip.so, as in bench.py."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_batch_entry_points_under_rccl_world1():
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "rccl_worker.py")
    p = subprocess.run([sys.executable, worker], capture_output=True, timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    assert b"RCCL_WORKER_OK" in p.stdout


def test_library_before_torch_keeps_torch_on_the_gpu():
    """Round 1's import-order trap: liblkhip.so loaded before torch left torch without a device.  _capi now loads
    torch's bundled HIP runtime first (one runtime per process), so either order works."""
    import subprocess
    import sys
    p = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "import_order_worker.py")], capture_output=True,
                       timeout=600)
    assert p.returncode == 0 and b"IMPORT_ORDER_OK" in p.stdout, p.stderr.decode()[-1500:]
