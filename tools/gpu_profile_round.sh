#!/bin/bash
# Runs on the GPU box (gpurun): rocprofv3 kernel traces and PMC passes of the round, summaries under gpurun_out/prof_r03/.
# PMC passes are their own runs with --kernel-trace only (no other trace domain beside --pmc).
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_r03
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
$R/tools/gpu_pmc_sq.sh 16384 $OUT/pmc_sq_e8m.txt > /dev/null 2>&1
$R/tools/gpu_pmc_sq.sh 256 $OUT/pmc_sq_c3.txt > /dev/null 2>&1
cd $R
hipcc --offload-arch=gfx950 -O3 tools/issue_rate.hip -o /tmp/issue_rate && /tmp/issue_rate > $OUT/valu_issue_rate.txt 2>&1
python tools/gpu_sweep.py 256 1024 4096 16384 32768 > $OUT/edge_sweep.txt 2>&1
(echo "# k_stream (float32 per edge, the default from 2048 tiles) against k_tile (float64 per edge) forced on the same graphs"; for M in 512 1024; do python tools/gpu_sweep.py $M; BT_STREAM_MIN_TILES=100000000 BT_EDGE_MIN_TILES=100000000 python tools/gpu_sweep.py $M; done) > $OUT/kernel_choice_f64.txt 2>&1
(for w in 2 4 6 8; do echo "== BT_EDGE_WAVES_PER_CU=$w"; BT_EDGE_WAVES_PER_CU=$w python tools/gpu_sweep.py 16384; done) > $OUT/edge_occupancy.txt 2>&1
python bench.py --steps 200 --warmup 20 > $OUT/bench_n1.json 2> $OUT/bench_n1.stderr
python tests/sequence_report.py > $OUT/sequence_ate.txt 2>&1
(echo "# 200 frames, steady state of the window"; python tests/sequence_report.py --frames 200 --skip-oracle; echo "# the same with BT_PLAN_SHIFT=0 (every plan from scratch)"; BT_PLAN_SHIFT=0 python tests/sequence_report.py --frames 200 --skip-oracle) >> $OUT/sequence_ate.txt 2>&1
(python tools/gpu_plan_time.py; python tools/gpu_plan_time.py window; BT_PLAN_PROF=1 python tools/gpu_plan_time.py window 2>&1 | tail -40) > $OUT/plan_time.txt 2>&1
python tools/gpu_check.py > $OUT/parity_numbers.txt 2>&1
python tools/gpu_refine_check.py >> $OUT/parity_numbers.txt 2>&1
python tools/gpu_edge_accuracy.py 64 256 >> $OUT/parity_numbers.txt 2>&1
(echo "# the float32 per-edge kernels (k_edge forced) on the same generated graphs"; BT_EDGE_MIN_TILES=1 python tools/gpu_edge_accuracy.py) >> $OUT/parity_numbers.txt 2>&1
(for w in C3 window; do python tools/gpu_timing.py --workload $w; BT_EDGE_PREC=0 BT_ETILE=0 python tools/gpu_timing.py --workload $w; BT_ETILE=0 python tools/gpu_timing.py --workload $w; done) > $OUT/timing_variants.txt 2>&1
ls -la $OUT
