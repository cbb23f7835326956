#!/bin/bash
# probe: does RCCL accept two ranks of one communicator on the same device (it would let the N > 1 RCCL path run on this one-GPU box)?
cd "$(dirname "$0")/.." || exit 1
export HPDDM_TEST_RCCL_SAME_GPU=1 NCCL_DEBUG=WARN
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1 --master-port 29711 tests/dist_worker.py rccl 2>&1 | grep -i "duplicate\|invalid usage\|HpddmHipError\|transport_rccl\|AssertionError\|DIST_WORKER_OK\|unhandled" | head -20
