#!/bin/bash
# GPU box: sample the shader clock while the 1M and the 100M bench run (are small-kernel steps clock-limited?)
mkdir -p gpurun_out
( python bench.py --steps 20000 --warmup 10 --no-extra --no-cpu-baseline > gpurun_out/clk_bench1m.log 2>&1 ) &
BP=$!
sleep 25
for i in 1 2 3 4 5 6; do rocm-smi --showclocks --showpower --showuse 2>/dev/null | grep -E "sclk|mclk|fclk|Power|busy" | tr '\n' ' '; echo; sleep 0.5; done
wait $BP
tail -1 gpurun_out/clk_bench1m.log | cut -c1-200
( python bench.py --particles 100000000 --gravity on --steps 300 --warmup 3 --no-extra --no-cpu-baseline > gpurun_out/clk_bench100m.log 2>&1 ) &
BP=$!
sleep 32
for i in 1 2 3 4; do rocm-smi --showclocks --showpower --showuse 2>/dev/null | grep -E "sclk|mclk|fclk|Power|busy" | tr '\n' ' '; echo; sleep 0.5; done
wait $BP
tail -1 gpurun_out/clk_bench100m.log | cut -c1-200
