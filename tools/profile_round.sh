#!/bin/bash
# rocprofv3 passes for profiles/ (run on the GPU box from the repo root):
#   tools/profile_round.sh ROUND TAG   e.g.  tools/profile_round.sh r3 b
# writes under gpurun_out/prof_ROUND_TAG/ (copy into profiles/ to commit):
#   ROUND_headline_kernel_stats_TAG.csv     --kernel-trace --stats of `bench.py --no-cpu-baseline --no-side-metrics`
#                                           (the headline workload ONLY: one symbol = one shape, B = 1024)
#   ROUND_pmc_fetch_size_TAG.csv, ROUND_pmc_write_size_TAG.csv, ROUND_pmc_traffic_TAG.json      (per-kernel, per launch)
#   ROUND_pmc_sq_TAG.csv                    SQ activity counters (MFMA busy, waits)
#   ROUND_acktr_kernel_stats_TAG.csv, ROUND_acktr_pmc_{fetch,write}_size_TAG.csv, ROUND_acktr_pmc_traffic_TAG.json
#                                           the same for the ACKTR update (tools/acktr_bench.py)
# Every --pmc pass is a run of its own (never combined with a trace domain), under its own time limit, with the program
# directly behind `--`; the script stops at the first failing step.
set -e
R=${1:-r3}
TAG=${2:-cur}
OUT=gpurun_out/prof_${R}_${TAG}
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-side-metrics"
ACKTR="python3 tools/acktr_bench.py"
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- $BENCH > $OUT/kt.log 2>&1
cp $(find $OUT/kt -name '*kernel_stats.csv' | head -1) $OUT/${R}_headline_kernel_stats_${TAG}.csv
cp $(find $OUT/kt -name '*kernel_trace.csv' | head -1) $OUT/kernel_trace.csv
python3 tools/trace_step.py $OUT/kernel_trace.csv > $OUT/${R}_headline_step_trace_${TAG}.txt || true
rm -rf $OUT/kt $OUT/kernel_trace.csv
for C in FETCH_SIZE WRITE_SIZE; do
    lc=$(echo $C | tr 'A-Z' 'a-z')
    timeout -s KILL 300 rocprofv3 --pmc $C --output-format csv -d $OUT/$C -o pmc -- $BENCH > $OUT/$C.log 2>&1
    python3 tools/pmc_summary.py $(find $OUT/$C -name '*counter_collection.csv' | head -1) $OUT/${R}_pmc_${lc}_${TAG}.csv
    rm -rf $OUT/$C
done
python3 tools/pmc_traffic.py $OUT/${R}_pmc_fetch_size_${TAG}.csv $OUT/${R}_pmc_write_size_${TAG}.csv $OUT/${R}_pmc_traffic_${TAG}.json
timeout -s KILL 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $OUT/SQ -o pmc -- $BENCH > $OUT/SQ.log 2>&1
python3 tools/pmc_summary.py $(find $OUT/SQ -name '*counter_collection.csv' | head -1) $OUT/${R}_pmc_sq_${TAG}.csv
rm -rf $OUT/SQ
# ---- the ACKTR update
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/akt -o kt -- $ACKTR > $OUT/akt.log 2>&1
cp $(find $OUT/akt -name '*kernel_stats.csv' | head -1) $OUT/${R}_acktr_kernel_stats_${TAG}.csv
rm -rf $OUT/akt
# (the counter passes keep every eigensolve on the hand-written Jacobi: rocprofv3's counter collection crashed inside a
#  dispatch issued from the host thread that runs rocSOLVER beside it)
export ACKTR_EIG_BLOCK_MAX_N=2048
for C in FETCH_SIZE WRITE_SIZE; do
    lc=$(echo $C | tr 'A-Z' 'a-z')
    timeout -s KILL 300 rocprofv3 --pmc $C --output-format csv -d $OUT/a$C -o pmc -- $ACKTR > $OUT/a$C.log 2>&1
    python3 tools/pmc_summary.py $(find $OUT/a$C -name '*counter_collection.csv' | head -1) $OUT/${R}_acktr_pmc_${lc}_${TAG}.csv
    rm -rf $OUT/a$C
done
python3 tools/pmc_traffic.py $OUT/${R}_acktr_pmc_fetch_size_${TAG}.csv $OUT/${R}_acktr_pmc_write_size_${TAG}.csv $OUT/${R}_acktr_pmc_traffic_${TAG}.json
echo profile_round done
