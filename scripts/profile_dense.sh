#!/bin/bash
# Round profiling recipe for the dense-metric path (GPU box): rocprofv3 kernel stats of a dense-metric run of the 2016
# posterior, then the HBM read counter in a pass of its own (no trace domains besides the kernel trace).
# Outputs under gpurun_out/prof_dense/; summaries are copied into profiles/ by hand.
chains=${1:-8}
iters=${2:-30}
out=$PWD/gpurun_out/prof_dense
mkdir -p $out
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $out/stats -o r -- python scripts/micro/dense_probe.py sampler $chains $iters > $out/run_stats.json 2> $out/stats.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $out/pmc_fetch -o r -- python scripts/micro/dense_probe.py sampler $chains 12 > $out/run_fetch.json 2> $out/fetch.err
python scripts/summarize_rocprof.py $(find $out -name "*_results.db" | sort) > $out/summary.txt 2>&1
cat $out/run_stats.json | cut -c1-600
cut -c1-200 $out/summary.txt | head -60
