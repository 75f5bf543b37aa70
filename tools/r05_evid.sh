#!/bin/bash
# round-5 evidence, part 1: the three profile passes of bench.py (kernel trace; FETCH_SIZE; WRITE_SIZE), then the SQ counters
tag=${1:-r05c}
tools/profile_bench.sh $tag || { echo "profile_bench failed"; tail -5 gpurun_out/${tag}_kt.err; exit 1; }
ls -la gpurun_out/${tag}_kernel_stats_bench_full.csv gpurun_out/${tag}_pmc_hbm_bytes.json gpurun_out/${tag}_kernel_trace_headline_kernels.txt
cat gpurun_out/${tag}_kernel_trace_headline_kernels.txt | cut -c1-140
