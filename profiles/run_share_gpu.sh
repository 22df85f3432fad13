#!/bin/bash
# Rehearsal of `bench.py --gpus N` (N > 1) on a ONE-GPU box: N ranks share the card.  Exercises the real kernels on the
# ranks' bands, HIP-event timing per rank, the gather / all_gather / all_reduce calls (over gloo: RCCL refuses two ranks on one card) and the JSON line.  The numbers
# are no measurement (the line says so and carries value = null).  Usage: bash profiles/run_share_gpu.sh N [workload]
set -o pipefail
N=${1:-2}; WL=${2:-c3}
export APT_BENCH_SHARE_GPU=1 HSA_ENABLE_IPC_MODE_LEGACY=0 NCCL_DEBUG=WARN
mkdir -p gpurun_out
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29533 \
    bench.py --gpus $N --steps 3 --warmup 1 --workload $WL > gpurun_out/share_gpu_$N.log 2>&1
rc=$?
tail -5 gpurun_out/share_gpu_$N.log
exit $rc
