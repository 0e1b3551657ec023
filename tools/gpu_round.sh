#!/bin/bash
# Run on the GPU box (via gpurun): GPU tests, smoke, bench, rocprofv3 kernel trace + PMC passes.
# Outputs land in gpurun_out/ (merged back by gpurun); summaries get copied to profiles/ by hand.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-round}
mkdir -p $OUT
cd $R
make -s -C oracle oracle
python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" >> $OUT/smoke.log
python bench.py --steps 20 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?" >> $OUT/bench.err
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats -o bench -- python $R/bench.py --steps 20 --warmup 3 --no-cpu --no-decode > $OUT/prof_stats.log 2>&1
# PMC passes, each on its own (no trace domains combined with --pmc)
rocprofv3 --output-format csv --pmc FETCH_SIZE -d $OUT/prof_pmc_fetch -o bench -- python $R/bench.py --steps 5 --warmup 1 --no-cpu --no-decode > $OUT/prof_pmc_fetch.log 2>&1
rocprofv3 --output-format csv --pmc WRITE_SIZE -d $OUT/prof_pmc_write -o bench -- python $R/bench.py --steps 5 --warmup 1 --no-cpu --no-decode > $OUT/prof_pmc_write.log 2>&1
rocprofv3 --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY -d $OUT/prof_pmc_sq -o bench -- python $R/bench.py --steps 5 --warmup 1 --no-cpu --no-decode > $OUT/prof_pmc_sq.log 2>&1
cd $R
# keep only the small summaries: gpurun merges back at most 64 MiB
python tools/prof_summarise.py $OUT > $OUT/prof_summary.txt 2>&1
find $OUT/prof_* -type f \( -name "*.db" -o -name "*.json" -o -name "*_kernel_trace.csv" -o -name "*_counter_collection.csv" -o -name "*.pftrace" -o -name "*agent_info.csv" \) -delete
du -sh $OUT; find $OUT -type f | head -40; cat $OUT/prof_summary.txt
tail -3 $OUT/pytest_gpu.log; tail -2 $OUT/smoke.log; cat $OUT/bench.json; tail -2 $OUT/bench.err
