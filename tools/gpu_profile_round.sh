#!/bin/bash
# Runs on the GPU box (gpurun): rocprofv3 kernel traces and PMC passes of the round, summaries under gpurun_out/prof_r06/
# (copied into profiles/r06_* afterwards).  PMC passes are their own runs with --kernel-trace only (no other trace domain
# beside --pmc).  profiles/pmc_k_tile.json is rewritten from the FETCH_SIZE / WRITE_SIZE passes of THIS build
# (tools/pmc_to_json.py: it carries the hash of the kernel sources; bench.py refuses it when the sources change).
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_r06
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
trace() {   # name, command...
    local name=$1; shift
    rm -rf /tmp/prof_$name
    rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -o runc -- "$@" > $OUT/$name.stdout 2> $OUT/$name.stderr
    python $R/tools/prof_summary.py $(find /tmp/prof_$name -name "*results.db" | head -1) > $OUT/kernel_trace_stats_$name.txt 2>&1
}
pmc() {     # counter, tag, M  (environment of the caller selects the kernels)
    rm -rf /tmp/pmc_$1_$2
    rocprofv3 --kernel-trace --pmc $1 -d /tmp/pmc_$1_$2 -o runc -- python $R/tools/gpu_pmc_run.py $3 4 > /dev/null 2> $OUT/pmc_$1_$2.stderr
    cp $(find /tmp/pmc_$1_$2 -name "*results.db" | head -1) /tmp/pmc_$1_$2.db
    python $R/tools/pmc_summary.py /tmp/pmc_$1_$2.db >> $OUT/pmc_fetch_write.txt 2>&1
}
# `tools/gpu_profile_round.sh pmc`: only the FETCH_SIZE / WRITE_SIZE passes and pmc_k_tile.json (after a late change of the kernel
# sources: the JSON carries their hash)
if [ "${1:-all}" = "pmc" ]; then
    : > $OUT/pmc_fetch_write.txt
    for C in FETCH_SIZE WRITE_SIZE; do pmc $C c3 256; pmc $C e8m 16384; BT_FORCE=wpt=0 pmc $C e8mf64 16384; done
    python $R/tools/pmc_to_json.py $OUT/pmc_k_tile.json \
        C3:131072:16384:64:/tmp/pmc_FETCH_SIZE_c3.db:/tmp/pmc_WRITE_SIZE_c3.db \
        E8M:8388608:1048576:64:/tmp/pmc_FETCH_SIZE_e8m.db:/tmp/pmc_WRITE_SIZE_e8m.db \
        E8M_float64_tile_kernels:8388608:1048576:64:/tmp/pmc_FETCH_SIZE_e8mf64.db:/tmp/pmc_WRITE_SIZE_e8mf64.db > $OUT/pmc_to_json.stdout 2>&1
    exit 0
fi
# ---- headline bench first (fresh clocks), then the traces
python $R/bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.stderr
trace c3 python $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-large
# `tools/gpu_profile_round.sh quick`: the bench line, the two traces the roofline figures rest on, the edge and ragged sweeps
if [ "${1:-all}" = "quick" ]; then
    trace e8m python $R/tools/gpu_pmc_run.py 16384 6
    cd $R
    python tools/gpu_sweep.py 256 4096 16384 32768 > $OUT/edge_sweep.txt 2>&1
    python tools/gpu_ragged_sweep.py > $OUT/ragged_graphs.txt 2>&1
    ls -la $OUT
    exit 0
fi
trace window python $R/tools/gpu_timing.py --workload window
trace e2m python $R/tools/gpu_pmc_run.py 4096 6
trace e8m python $R/tools/gpu_pmc_run.py 16384 6
BT_FORCE=wpt=0 trace e8m_f64tile python $R/tools/gpu_pmc_run.py 16384 6
trace ga python $R/tools/gpu_ga_bench.py
# ---- HBM traffic of the Jacobian kernel: C3 (k_tile, float64) and 8.4M edges (k_edge2, mixed precision: the default there; and the
# float64 tile kernel the caller can have instead)
: > $OUT/pmc_fetch_write.txt
for C in FETCH_SIZE WRITE_SIZE; do pmc $C c3 256; pmc $C e8m 16384; BT_FORCE=wpt=0 pmc $C e8mf64 16384; done
# the same counters with L2 and MALL flushed between the steps (a 512 MB sweep: BT_PMC_COLD=1), and the kernel times warm / cold
echo "# ---- cold: a 512 MB sweep between the steps (BT_PMC_COLD=1)" >> $OUT/pmc_fetch_write.txt
for C in FETCH_SIZE WRITE_SIZE; do BT_PMC_COLD=1 pmc $C c3cold 256; BT_PMC_COLD=1 pmc $C e8mcold 16384; done
python - >> $OUT/pmc_fetch_write.txt 2>&1 <<PYEOF
import json
b = json.loads(open("$OUT/bench_n1.json").read().strip().splitlines()[-1])
r = b["roofline"]
print("# ---- Jacobian kernel, warm (steps back to back) / cold (512 MB sweep between launches), HIP events (bench.py)")
print(f"C3      {r['kernel']}: warm {r.get('kernel_us')} us, cold {r.get('cold_kernel_us')} us (frac {r.get('frac')} / {r.get('cold_frac')})")
w = b["config"].get("sliding_window", {})
print(f"window  {w.get('jacobian_kernel')}: warm {w.get('kernel_us', {}).get('tile')} us, cold {w.get('cold_kernel_us_tile')} us")
l = r.get("large", {})
print(f"8.4M    {l.get('kernel')}: warm {l.get('kernel_us')} us, cold {l.get('cold_kernel_us')} us (frac {l.get('frac')} / {l.get('cold_frac')})")
PYEOF
python $R/tools/pmc_to_json.py $OUT/pmc_k_tile.json \
    C3:131072:16384:64:/tmp/pmc_FETCH_SIZE_c3.db:/tmp/pmc_WRITE_SIZE_c3.db \
    E8M:8388608:1048576:64:/tmp/pmc_FETCH_SIZE_e8m.db:/tmp/pmc_WRITE_SIZE_e8m.db \
    E8M_float64_tile_kernels:8388608:1048576:64:/tmp/pmc_FETCH_SIZE_e8mf64.db:/tmp/pmc_WRITE_SIZE_e8mf64.db > $OUT/pmc_to_json.stdout 2>&1
$R/tools/gpu_pmc_sq.sh 256 $OUT/pmc_sq_c3.txt > /dev/null 2>&1
$R/tools/gpu_pmc_sq.sh 16384 $OUT/pmc_sq_e8m.txt > /dev/null 2>&1
cd $R
# ---- the edge sweep in both precisions (the roofline table of DESIGN.md §6)
(echo "# default: k_tile (float64 per edge) below 2048 tiles, k_stream / k_edge (mixed precision) from there: every row inside the 1e-5 bar on the update"; python tools/gpu_sweep.py 256 1024 4096 16384 32768;
 echo "# BT_FORCE=wpt=0: the float64 tile kernel at every size"; BT_FORCE=wpt=0 python tools/gpu_sweep.py 4096 16384 32768) > $OUT/edge_sweep.txt 2>&1
# ---- solver: k_solve_pipe with its in-kernel cycle counters (k_solve_chain, round 4's equal-time alternative, is deleted)
(for w in C3 window; do python tools/gpu_timing.py --workload $w; done;
 BT_DEBUG_MODE=16 python tools/gpu_timing.py | grep -A2 "solver"; BT_FORCE=solver=fused BT_DEBUG_MODE=16 python tools/gpu_timing.py | grep -A2 "solver";
 BT_DEBUG_MODE=16 python tools/gpu_timing.py --workload window | grep -A2 "solver") > $OUT/solver_variants.txt 2>&1
# ---- k_edge2 forced from 2048 tiles on (default: from 4096), and the window kernel with one / two rounds per trip
# (profiles/r05_edge2_vs_edge.txt also holds round 4's k_edge on the same graphs: measured before its pose+structure
#  instantiation was removed, at the commit named in the file)
(echo "# k_edge2 forced from 2048 tiles on (BT_FORCE=kernel=k_edge2)"; BT_FORCE=kernel=k_edge2 python tools/gpu_sweep.py 2048 4096 8192 16384 32768) 2>&1 | cut -c1-330 > $OUT/edge2_forced.txt
(python tools/gpu_timing.py --workload window) > $OUT/window_rounds_per_trip.txt 2>&1
python tools/gpu_spec_time.py > $OUT/plan_call_host_time.txt 2>&1
(for i in 1 2 3; do AB=1 FRAMES=400 python tools/gpu_update_floor.py; done; BT_PLAN_PRESHIFT=0 python tools/gpu_update_floor.py | tail -2; PREFETCH=1 python tools/gpu_update_floor.py | tail -2) 2>&1 | grep -v amdgpu.ids > $OUT/update_floor.txt
python tests/sequence_report.py > $OUT/sequence_ate.txt 2>&1
(echo "# 200 frames, steady state of the window"; python tests/sequence_report.py --frames 200 --skip-oracle) >> $OUT/sequence_ate.txt 2>&1
(python tools/gpu_plan_time.py; python tools/gpu_plan_time.py window; python tools/gpu_plan_time.py large) > $OUT/plan_time.txt 2>&1
python tools/gpu_check.py > $OUT/parity_numbers.txt 2>&1
python tools/gpu_refine_check.py >> $OUT/parity_numbers.txt 2>&1
python tools/gpu_edge_accuracy.py 64 256 >> $OUT/parity_numbers.txt 2>&1
(echo "# graphs of 2048 .. 8192 tiles, default (k_stream / k_edge, mixed precision)"; python tools/gpu_edge_accuracy.py 2048 8192;
 echo "# the same with the float64 tile kernel (BT_FORCE=wpt=0)"; BT_FORCE=wpt=0 python tools/gpu_edge_accuracy.py 2048 8192) >> $OUT/parity_numbers.txt 2>&1
(for w in C3 window; do python tools/gpu_timing.py --workload $w; BT_FORCE=kernel=k_tile,prec=f32 python tools/gpu_timing.py --workload $w; BT_FORCE=kernel=k_tile python tools/gpu_timing.py --workload $w; done) > $OUT/timing_variants.txt 2>&1
python tools/gpu_ga_bench.py > $OUT/global_refine_losses.txt 2>&1
python tools/gpu_ragged_sweep.py > $OUT/ragged_graphs.txt 2>&1
# ---- N > 1 plumbing on the one GPU (ranks share it; gloo rendezvous): not a scaling measurement
for n in 2 4 8; do
  BT_BENCH_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29820 + n)) bench.py --gpus $n --steps 50 --warmup 5 2> $OUT/bench_plumbing_n$n.stderr | tail -1 > $OUT/bench_plumbing_n$n.json
done
ls -la $OUT
