#!/bin/bash
# The N > 1 path of bench.py end to end on a ONE-GPU box (correctness only, not a measurement): two ranks pinned to GPU 0,
# the slice exchange and the partial-sum all-gather over gloo through host memory.  The driver's real run uses one rank per
# GPU and RCCL ("nccl") over xGMI.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/rehearsal
G16_BENCH_DEVICE=0 G16_BENCH_BACKEND=gloo timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
  --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 5 --warmup 2 > gpurun_out/rehearsal/n2.json 2> gpurun_out/rehearsal/n2.err
echo "rc=$?"; tail -3 gpurun_out/rehearsal/n2.err; tail -c 1500 gpurun_out/rehearsal/n2.json
