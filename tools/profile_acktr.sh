#!/bin/bash
# rocprofv3 passes of the ACKTR update (configs[3]: NatureCNN, 32 procs x 20 steps, Ts=1, Tf=10) for profiles/:
#   tools/profile_acktr.sh TAG  ->  gpurun_out/prof_acktr_TAG/{kernel_stats.csv, pmc_FETCH_SIZE.csv, pmc_WRITE_SIZE.csv}
# One --kernel-trace --stats pass, then FETCH_SIZE and WRITE_SIZE in --pmc passes of their own (never combined with a
# trace domain), each under its own time limit; stops at the first failing step.
set -e
TAG=${1:-cur}
OUT=gpurun_out/prof_acktr_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
PROG="python3 tools/acktr_bench.py"
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- $PROG > $OUT/kt.log 2>&1
cp $(find $OUT/kt -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats.csv
# (the counter passes keep every eigensolve on the hand-written Jacobi: rocprofv3's counter collection crashed inside a
#  dispatch issued from the host thread that runs rocSOLVER beside it)
export ACKTR_EIG_BLOCK_MAX_N=2048
for C in FETCH_SIZE WRITE_SIZE; do
    timeout -s KILL 300 rocprofv3 --pmc $C --output-format csv -d $OUT/$C -o pmc -- $PROG > $OUT/$C.log 2>&1
    python3 tools/pmc_summary.py $(find $OUT/$C -name '*counter_collection.csv' | head -1) $OUT/pmc_$C.csv
    rm -rf $OUT/$C
done
rm -rf $OUT/kt
tail -2 $OUT/kt.log
