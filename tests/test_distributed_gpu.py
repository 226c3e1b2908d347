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


def test_device_resident_gather_under_rccl_world2():
    """Two ranks, each driving its own GPU with ALL devices visible (the torchrun situation of `bench.py --gpus N`): the
    device-resident all-gather of lombscargle_batch and the 16-B-per-target gather of lombscargle_peaks_batch.  Needs two
    GPUs: skipped on the 1-GPU box (the gloo world-2 tests cover the sharding logic there)."""
    import socket
    from lightkurve_amd import _capi
    if _capi.device_count() < 2:
        pytest.skip("needs >= 2 GPUs")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = str(s.getsockname()[1])
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "rccl_worker.py")
    env = {k: v for k, v in os.environ.items() if k not in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES")}
    procs = [subprocess.Popen([sys.executable, worker, str(r), "2", port], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
             for r in range(2)]
    for r, p in enumerate(procs):
        out, err = p.communicate(timeout=600)
        assert p.returncode == 0 and b"RCCL_WORKER_OK" in out, (r, err.decode()[-2000:])


def test_library_before_torch_keeps_torch_on_the_gpu():
    """Round 1's import-order trap: liblkhip.so loaded before torch left torch without a device.  _capi now loads
    torch's bundled HIP runtime first (one runtime per process), so either order works."""
    import subprocess
    import sys
    p = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "import_order_worker.py")], capture_output=True,
                       timeout=600)
    assert p.returncode == 0 and b"IMPORT_ORDER_OK" in p.stdout, p.stderr.decode()[-1500:]
