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


FAKE_RCCL = os.path.join(HERE, "fake_rccl", "libfake_rccl.so")


def _fake_rccl_env(host):
    """tests/fake_rccl: a double of librccl over /dev/shm bound through HPDDM_HIP_RCCL_LIB (built by __graft_entry__.build())"""
    if not os.path.exists(FAKE_RCCL):
        subprocess.check_call(["make", "-C", os.path.dirname(FAKE_RCCL)])
    return {"HPDDM_HIP_RCCL_LIB": FAKE_RCCL, "FAKE_RCCL_HOST": "1" if host else "0", "HPDDM_TEST_RCCL_SAME_GPU": "1"}


def _launch(mode, world, port, *extra, env_extra=None):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(HERE, "dist_worker.py"), mode, *extra]
    env = dict(os.environ, OMP_NUM_THREADS="1" if world > 4 else "2", HPDDM_HIP_NUM_THREADS="1" if world > 4 else "2")
    env.update(env_extra or {})
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert res.returncode == 0 and "DIST_WORKER_OK" in res.stdout, res.stdout[-3000:] + res.stderr[-3000:]


@pytest.mark.parametrize("world", [2, 3, 8])
def test_halo_lists_gloo_cpu(world):
    """world 8 = the topology of configs[3]: 4 x 4 x 4 subdomains in 2 x 2 x 2 bricks, one brick per rank, 7 peers each"""
    _launch("lists", world, 29620 + world)


@pytest.mark.parametrize("world", [2, 8])
def test_rccl_transport_sequence_cpu(world):
    """the product's RcclTransport (grouped ncclSend / ncclRecv with the partition's peer offsets, ncclSum, ncclMax) driven on the CPU
    against the librccl double: every value arrives where the other end of its link put it, sizes agree on both ends of every link"""
    _launch("lists_rccl", world, 29660 + world, env_extra=_fake_rccl_env(host=True))


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 4, 8])
def test_sharded_operator_rccl_transport_shared_gpu(world):
    """the whole sharded operator (one- and two-level apply, GMV, GMRES, coarse assembly across the ranks, the three norms of
    computeResidual) through the library's OWN RcclTransport -- device buffers, communication stream, events -- with the ranks sharing
    GPU 0: librccl itself refuses two ranks on one device, its double over /dev/shm does not.  world 8 = the layout of configs[3]."""
    _launch("rccl", world, 29670 + world, env_extra=_fake_rccl_env(host=False))


@pytest.mark.gpu
def test_sharded_complex_operator_rccl_transport_shared_gpu():
    _launch("rccl", 4, 29680, "helmholtz", env_extra=_fake_rccl_env(host=False))


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3, 4])
def test_sharded_operator_shared_gpu(world):
    """world 4 = the layout of configs[4]: 2 x 4 x 4 subdomains, one 2 x 2 x 2 brick per rank"""
    _launch("gpu", world, 29630 + world)


@pytest.mark.gpu
def test_sharded_complex_operator_shared_gpu():
    """configs[4] in small on its own layout: complex<double> operator on 4 ranks (32 subdomains), two-level, Block GMRES"""
    _launch("gpu", 4, 29650, "helmholtz")


@pytest.mark.gpu
def test_rccl_binding_one_rank():
    from hpddm_amd import hpddm
    hpddm.require_device()
    hpddm.rccl_self_test()


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 4, 8])
def test_sharded_operator_native_rccl(world):
    from hpddm_amd import hpddm
    if hpddm.device_count() < world:
        pytest.skip(f"native RCCL transport across {world} ranks needs {world} GPUs (one process per GPU)")
    _launch("rccl", world, 29640 + world)
