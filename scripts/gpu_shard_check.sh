#!/bin/bash
# GPU box: sharded parity tests (both exchanges), then the N-rank bench rehearsal on one GPU (gloo staging).
set -u
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_sharded.py -q -x --timeout 500 > gpurun_out/shard_tests.log 2>&1
rc=$?; tail -15 gpurun_out/shard_tests.log; echo "shard tests rc=$rc"
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
for N in ${RANKS:-2 4}; do
  GPE_BENCH_BACKEND=gloo GPE_BENCH_SHARE_GPU=1 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29520+N)) bench.py --gpus $N --steps ${STEPS:-100} --warmup 10 --no-extra > gpurun_out/shard_bench_$N.log 2>&1
  rc=$?; tail -4 gpurun_out/shard_bench_$N.log | cut -c1-1500; echo "bench $N rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
done
