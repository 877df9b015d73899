# instruction-cache counters of the cluster kernel (development aid)
export TMPDIR=/tmp
out=$PWD/gpurun_out/icache
mkdir -p $out
POTUS_K=16 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES --kernel-trace -d $out/a -o r -- python scripts/gpu_probe.py 60 8 > $out/a.log 2>&1
POTUS_K=16 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_IFETCH --kernel-trace -d $out/b -o r -- python scripts/gpu_probe.py 60 8 > $out/b.log 2>&1
python scripts/summarize_rocprof.py $(find $out -name "*_results.db" | sort) 2>&1 | grep -v "^$" | cut -c1-200
tail -3 $out/a.log $out/b.log
