#!/bin/bash
# only the PMC passes of tools/gpu_round3.sh (128-lane engine) -> gpurun_out/<name>/pmc_traffic.json
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; NAME=${1:-pmc}; OUT=$R/gpurun_out/$NAME; mkdir -p $OUT; cd $R
make -s -C oracle oracle
export TMPDIR=/tmp S3A_BENCH_NO_RCCL=1 ${PMC_ENV:-}
cd /tmp
PMC_ARGS="--steps 1 --warmup 0 --plain --frames 100 --utts 64 --lanes 64 --engines 1"
rocprofv3 --output-format csv --pmc FETCH_SIZE -d $OUT/prof_pmc_fetch -o bench -- python $R/bench.py $PMC_ARGS > $OUT/prof_pmc_fetch.log 2>&1; echo "fetch rc=$?"
rocprofv3 --output-format csv --pmc WRITE_SIZE -d $OUT/prof_pmc_write -o bench -- python $R/bench.py $PMC_ARGS > $OUT/prof_pmc_write.log 2>&1; echo "write rc=$?"
timeout 600 rocprofv3 --output-format csv --pmc FETCH_SIZE -d $OUT/prof_scoring_pmc_fetch -o scoring -- python $R/bench.py --only-scoring > $OUT/prof_scoring_pmc_fetch.log 2>&1
timeout 600 rocprofv3 --output-format csv --pmc WRITE_SIZE -d $OUT/prof_scoring_pmc_write -o scoring -- python $R/bench.py --only-scoring > $OUT/prof_scoring_pmc_write.log 2>&1
cd $R
python tools/pmc_traffic.py $OUT $OUT/pmc_traffic.json "$NAME: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py $PMC_ARGS (a 64-lane engine: rocprofv3 counter collection crashes on the 128-lane launches)" 64 | tail -30
python tools/prof_summarise.py $OUT > $OUT/prof_summary.txt 2>&1
find $OUT/prof_* -type f \( -name "*.db" -o -name "*.json" -o -name "*_kernel_trace.csv" -o -name "*_counter_collection.csv" -o -name "*.pftrace" -o -name "*agent_info.csv" \) -delete
tail -5 $OUT/prof_pmc_fetch.log | cut -c1-200
