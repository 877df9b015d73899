#!/bin/bash
# Instruction-fetch counters of the bench command's sampler kernel: separate --pmc passes,
# kernel trace only.  usage: scripts/profile_icache.sh <tag> [steps]; outputs under gpurun_out/icache_<tag>/
tag=${1:-r06}
steps=${2:-4}
out=$PWD/gpurun_out/icache_$tag
mkdir -p $out
export TMPDIR=/tmp
rocprofv3 -L > $out/counters.txt 2>&1
grep -i -E "ICACHE|IFETCH|SQ_WAIT_INST|SQ_INST_LEVEL|SQC_" $out/counters.txt | head -60 > $out/counters_ifetch.txt
B="python bench.py --steps $steps --no-cpu-baseline --no-saturated --no-side"
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE --kernel-trace -d $out/p1 -o r -- $B > $out/bench_p1.json 2> $out/p1.err
rocprofv3 --pmc SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_WAVES --kernel-trace -d $out/p2 -o r -- $B > $out/bench_p2.json 2> $out/p2.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU --kernel-trace -d $out/p3 -o r -- $B > $out/bench_p3.json 2> $out/p3.err
python scripts/summarize_pmc.py $out > $out/summary.txt 2>&1
cat $out/summary.txt | cut -c1-250
