#!/bin/bash
# Runs on the GPU box (gpurun): rocprofv3 kernel traces and PMC passes of the round, summaries under gpurun_out/prof_r02/.
# PMC passes are their own runs with --kernel-trace only (no other trace domain beside --pmc).
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_r02
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
trace() {   # name, command...
    local name=$1; shift
    rm -rf /tmp/prof_$name
    rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -o runc -- "$@" > $OUT/$name.stdout 2> $OUT/$name.stderr
    python $R/tools/prof_summary.py $(find /tmp/prof_$name -name "*results.db" | head -1) > $OUT/kernel_trace_stats_$name.txt 2>&1
}
pmc() {     # counter, M
    rm -rf /tmp/pmc_$1_$2
    rocprofv3 --kernel-trace --pmc $1 -d /tmp/pmc_$1_$2 -o runc -- python $R/tools/gpu_pmc_run.py $2 4 > /dev/null 2> $OUT/pmc_$1_$2.stderr
    python $R/tools/pmc_summary.py $(find /tmp/pmc_$1_$2 -name "*results.db" | head -1) >> $OUT/pmc_fetch_write.txt 2>&1
}
trace c3 python $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline
trace window python $R/tools/gpu_timing.py --workload window
trace e8m python $R/tools/gpu_pmc_run.py 16384 6
: > $OUT/pmc_fetch_write.txt
for M in 256 16384; do for C in FETCH_SIZE WRITE_SIZE; do pmc $C $M; done; done
cd $R
python tools/gpu_sweep.py 256 1024 4096 16384 32768 > $OUT/edge_sweep.txt 2>&1
python bench.py --steps 200 --warmup 20 > $OUT/bench_n1.json 2> $OUT/bench_n1.stderr
python tests/sequence_report.py > $OUT/sequence_ate.txt 2>&1
python tools/gpu_ga_bench.py > $OUT/global_refine_losses.txt 2>&1
python tools/gpu_check.py > $OUT/parity_numbers.txt 2>&1
python tools/gpu_refine_check.py >> $OUT/parity_numbers.txt 2>&1
BT_EDGE_MIN_TILES=1 python tools/gpu_edge_accuracy.py >> $OUT/parity_numbers.txt 2>&1
ls -la $OUT
