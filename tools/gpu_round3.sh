#!/bin/bash
# Round-3 validation on the GPU box (via gpurun): all gpu tests, smoke, the default bench (CPU legs included), rocprofv3
# kernel traces of the bench, and the HBM-traffic PMC passes (each on its own: no trace domain next to --pmc) of a
# 128-lane engine -- the launch shape bench.py's roofline describes.  Outputs land in gpurun_out/<name>/ (merged back by
# gpurun); tools/pmc_traffic.py turns the PMC CSVs into profiles/pmc_traffic.json; summaries are copied to profiles/.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
NAME=${1:-r3}
OUT=$R/gpurun_out/$NAME
mkdir -p $OUT
cd $R
make -s -C oracle oracle
export S3A_ON_GPU_BOX=1
if [ "${SKIP_TESTS:-0}" != "1" ]; then
python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" >> $OUT/smoke.log
fi
( time python bench.py ${BENCH_ARGS:-} > $OUT/bench.json 2> $OUT/bench.err ) 2> $OUT/bench.time; echo "bench rc=$?" >> $OUT/bench.err
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats -o bench -- python $R/bench.py --steps 1 --warmup 1 --no-cpu --no-scoring > $OUT/prof_stats.log 2>&1
export S3A_BENCH_NO_RCCL=1
PMC_ARGS="--steps 1 --warmup 0 --no-cpu --no-scoring --frames 100 --utts 64 --lanes 64 --engines 1"
rocprofv3 --output-format csv --pmc FETCH_SIZE -d $OUT/prof_pmc_fetch -o bench -- python $R/bench.py $PMC_ARGS > $OUT/prof_pmc_fetch.log 2>&1
rocprofv3 --output-format csv --pmc WRITE_SIZE -d $OUT/prof_pmc_write -o bench -- python $R/bench.py $PMC_ARGS > $OUT/prof_pmc_write.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_pmc_stats -o bench -- python $R/bench.py $PMC_ARGS > $OUT/prof_pmc_stats.log 2>&1
# the scoring kernels alone (whole-utterance + frame-synchronous, hub4 and WSJ shapes): kernel trace and the two PMC passes
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_scoring_stats -o scoring -- python $R/bench.py --only-scoring > $OUT/scoring.json 2> $OUT/prof_scoring_stats.log
timeout 600 rocprofv3 --output-format csv --pmc FETCH_SIZE -d $OUT/prof_scoring_pmc_fetch -o scoring -- python $R/bench.py --only-scoring > $OUT/prof_scoring_pmc_fetch.log 2>&1
timeout 600 rocprofv3 --output-format csv --pmc WRITE_SIZE -d $OUT/prof_scoring_pmc_write -o scoring -- python $R/bench.py --only-scoring > $OUT/prof_scoring_pmc_write.log 2>&1
cd $R
python tools/prof_summarise.py $OUT > $OUT/prof_summary.txt 2>&1
python tools/pmc_traffic.py $OUT $OUT/pmc_traffic.json "$NAME: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py $PMC_ARGS (a 64-lane engine: rocprofv3 counter collection crashes on the 128-lane launches)" 64 > $OUT/pmc_traffic.log 2>&1
f=$(find $OUT/prof_stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/bench_kernel_stats.csv
f=$(find $OUT/prof_pmc_stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/engine128_kernel_stats.csv
f=$(find $OUT/prof_scoring_stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/scoring_kernel_stats.csv
find $OUT/prof_* -type f \( -name "*.db" -o -name "*.json" -o -name "*_kernel_trace.csv" -o -name "*_counter_collection.csv" -o -name "*.pftrace" -o -name "*agent_info.csv" \) -delete
du -sh $OUT; tail -3 $OUT/pytest_gpu.log 2>/dev/null; tail -2 $OUT/smoke.log 2>/dev/null; tail -3 $OUT/bench.err; cat $OUT/bench.time; head -c 600 $OUT/bench.json; echo; cat $OUT/pmc_traffic.log | tail -20; head -30 $OUT/prof_summary.txt
