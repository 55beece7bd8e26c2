#!/bin/bash
# Round 4, VERDICT task 2: the fixed cost of the sharded step loop on ONE GPU (one-rank RCCL communicator, the
# neighbour is the rank itself): wall time per step against the plain loop, then the rocprofv3 kernel stats of the
# same script -> gpurun_out/r04/shard_one_rank_*.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r04; mkdir -p $OUT
N=${1:-1000000}
cd $ROOT
timeout -k 10 300 python scripts/shard_overhead_one_rank.py $N 2000 > $OUT/shard_one_rank_overhead.txt 2>&1 || { echo "failed"; tail -20 $OUT/shard_one_rank_overhead.txt; exit 1; }
cat $OUT/shard_one_rank_overhead.txt
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/shard_trace; mkdir -p $OUT/shard_trace
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/shard_trace -- python3 $ROOT/scripts/shard_overhead_one_rank.py $N 400 > $OUT/shard_one_rank_trace.log 2>&1 || { echo "trace failed"; tail -5 $OUT/shard_one_rank_trace.log; exit 1; }
find $OUT/shard_trace -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} $OUT/shard_one_rank_kernel_stats.csv
f=$(find $OUT/shard_trace -name '*kernel_trace.csv' | head -1)
python3 $ROOT/scripts/trace_gaps.py $f > $OUT/shard_one_rank_gaps.txt 2>&1 || true
rm -rf $OUT/shard_trace
head -25 $OUT/shard_one_rank_kernel_stats.csv
tail -30 $OUT/shard_one_rank_gaps.txt
