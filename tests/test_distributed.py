"""The N>1 path (subdomains sharded over several GPUs) with world_size-2/3 multi-process runs:
CPU: the library's halo lists under a real gloo transport against the oracle's global exchange;
GPU: the full sharded operator (two processes sharing GPU 0, host-staged transport) against the oracle; with one GPU per rank
(skipped on a single-GPU box) the same checks through the library's own RCCL transport; on one GPU the RCCL binding itself
(one-rank communicator, grouped send/recv to self, all-reduce on the library stream)."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _launch(mode, world, port):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(HERE, "dist_worker.py"), mode]
    env = dict(os.environ, OMP_NUM_THREADS="2", HPDDM_HIP_NUM_THREADS="2")
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert res.returncode == 0 and "DIST_WORKER_OK" in res.stdout, res.stdout[-3000:] + res.stderr[-3000:]


@pytest.mark.parametrize("world", [2, 3])
def test_halo_lists_gloo_cpu(world):
    _launch("lists", world, 29620 + world)


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3])
def test_sharded_operator_shared_gpu(world):
    _launch("gpu", world, 29630 + world)


@pytest.mark.gpu
def test_rccl_binding_one_rank():
    from hpddm_amd import hpddm
    hpddm.require_device()
    hpddm.rccl_self_test()


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3])
def test_sharded_operator_native_rccl(world):
    from hpddm_amd import hpddm
    if hpddm.device_count() < world:
        pytest.skip(f"native RCCL transport across {world} ranks needs {world} GPUs (one process per GPU)")
    _launch("rccl", world, 29640 + world)
