#!/bin/bash
# rocprofv3 passes of the bench command for profiles/ (run on the GPU box from the repo root):
#   tools/profile_round.sh TAG     ->  gpurun_out/prof_TAG/{kernel_stats.csv, pmc_fetch.csv, pmc_write.csv, pmc_traffic.json}
# One --kernel-trace --stats pass, then FETCH_SIZE and WRITE_SIZE in --pmc passes of their own (never combined with
# a trace domain), each under its own time limit; the script stops at the first failing step.
set -e
TAG=${1:-cur}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-side-metrics"
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- $BENCH > $OUT/kt.log 2>&1
cp $(find $OUT/kt -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats.csv
for C in FETCH_SIZE WRITE_SIZE; do
    timeout -s KILL 300 rocprofv3 --pmc $C --output-format csv -d $OUT/$C -o pmc -- $BENCH > $OUT/$C.log 2>&1
    python3 tools/pmc_summary.py $(find $OUT/$C -name '*counter_collection.csv' | head -1) $OUT/pmc_$C.csv
    rm -rf $OUT/$C
done
rm -rf $OUT/kt
python3 tools/pmc_traffic.py $OUT/pmc_FETCH_SIZE.csv $OUT/pmc_WRITE_SIZE.csv $OUT/pmc_traffic.json
# optional third argument "sq": one more pass with the SQ activity counters (7 SQ_ + 1 GRBM_, inside the per-block limits)
if [ "$2" = "sq" ]; then
    timeout -s KILL 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $OUT/SQ -o pmc -- $BENCH > $OUT/SQ.log 2>&1
    python3 tools/pmc_summary.py $(find $OUT/SQ -name '*counter_collection.csv' | head -1) $OUT/pmc_SQ.csv
    rm -rf $OUT/SQ
fi
