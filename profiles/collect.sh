#!/bin/bash
# Collect the per-kernel evidence for one round on the GPU box:
#   profiles/collect.sh <tag>      (run from the repository root, through gpurun)
# writes gpurun_out/<tag>/{bench.json, stats/, pmc_fetch/, pmc_write/, pmc_sq/, pipeline_overlap.txt}.
# Each GPU step runs under its own time limit and the chain stops at the first failure.
# Counter passes run alone (kernel trace + stats only in the timing pass).
set -o pipefail
tag=${1:-prof}
out=gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
export MJ_BENCH_CACHE=/tmp/mj_bench_cache     # the passes below decode the same generated files
# the profiled command: the benchmark's timed region alone -- no CPU leg, no forked input
# generation (the profiler preload does not survive forks), no single-image and no
# pinned-host legs, whose launches of the same kernels on other sizes, or overlapping each
# other, would be averaged into the per-kernel figures
B="bench.py --steps 20 --warmup 3 --no-cpu-baseline --gen-workers 1 --no-singles --no-pipeline --parity none"
# the pinned-host pipeline for the copy / kernel timeline: the same flags with the pipelined region on
P="bench.py --steps 20 --warmup 3 --no-cpu-baseline --gen-workers 1 --no-singles --parity none --sustain-seconds 0"
timeout -k 10 300 python3 -u bench.py --steps 20 --warmup 3 > "$out/bench.json" 2> "$out/bench.err" &&
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -o run -- python3 $B > "$out/stats.log" 2>&1 &&
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/pmc_fetch" -o run -- python3 $B > "$out/pmc_fetch.log" 2>&1 &&
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/pmc_write" -o run -- python3 $B > "$out/pmc_write.log" 2>&1 &&
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d "$out/pmc_sq" -o run -- python3 $B > "$out/pmc_sq.log" 2>&1
rc=$?
if [ $rc -eq 0 ]; then
  timeout -k 10 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d "$out/pipe" -o run -- python3 $P > "$out/pipe.log" 2>&1 &&
  python3 tools/overlap.py "$out/pipe" 20 "$(python3 -c "import json,sys; print(json.loads([l for l in open('$out/pipe.log') if l.startswith('{')][-1])['config']['compressed_MB_per_gpu'])")" > "$out/pipeline_overlap.txt" 2>&1
  rc=$?
  find "$out/pipe" -name "*_memory_copy_trace.csv" -delete
fi
python3 profiles/pmc_summary.py "$out/pmc_fetch" "$out/pmc_write" "$out/pmc_sq" > "$out/pmc_summary.txt" 2>&1
find "$out" -name "*_kernel_trace.csv" -delete    # per-dispatch rows: large, the stats file has the averages
find "$out" -name "*_counter_collection.csv" -delete
echo "collect rc=$rc"
ls -R "$out" | head -40
exit $rc
