#!/bin/bash
# Round-2 validation on the GPU box (via gpurun): all gpu tests, smoke, the default bench, rocprofv3 kernel traces of
# the bench and of the wide-beam task, and the HBM-traffic PMC passes (each on its own: no trace domain next to --pmc).
# Outputs land in gpurun_out/<name>/ (merged back by gpurun); the summaries are copied to profiles/ by hand.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-r2}
mkdir -p $OUT
cd $R
make -s -C oracle oracle
export S3A_ON_GPU_BOX=1
python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" >> $OUT/smoke.log
python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?" >> $OUT/bench.err
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats -o bench -- python $R/bench.py --steps 1 --warmup 1 --no-cpu --no-scoring --lanes 256 > $OUT/prof_stats.log 2>&1
rocprofv3 --output-format csv --pmc FETCH_SIZE -d $OUT/prof_pmc_fetch -o bench -- python $R/bench.py --steps 1 --warmup 0 --no-cpu --no-scoring --frames 100 --utts 32 --lanes 64 --engines 1 > $OUT/prof_pmc_fetch.log 2>&1
rocprofv3 --output-format csv --pmc WRITE_SIZE -d $OUT/prof_pmc_write -o bench -- python $R/bench.py --steps 1 --warmup 0 --no-cpu --no-scoring --frames 100 --utts 32 --lanes 64 --engines 1 > $OUT/prof_pmc_write.log 2>&1
# the scoring kernels alone (whole-utterance + frame-synchronous, hub4 and WSJ shapes): kernel trace and the two PMC passes
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_scoring_stats -o scoring -- python $R/bench.py --only-scoring > $OUT/scoring.json 2> $OUT/prof_scoring_stats.log
timeout 600 rocprofv3 --output-format csv --pmc FETCH_SIZE -d $OUT/prof_scoring_pmc_fetch -o scoring -- python $R/bench.py --only-scoring > $OUT/prof_scoring_pmc_fetch.log 2>&1
timeout 600 rocprofv3 --output-format csv --pmc WRITE_SIZE -d $OUT/prof_scoring_pmc_write -o scoring -- python $R/bench.py --only-scoring > $OUT/prof_scoring_pmc_write.log 2>&1
cd $R
python tools/prof_summarise.py $OUT > $OUT/prof_summary.txt 2>&1
f=$(find $OUT/prof_stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/bench_kernel_stats.csv
find $OUT/prof_* -type f \( -name "*.db" -o -name "*.json" -o -name "*_kernel_trace.csv" -o -name "*_counter_collection.csv" -o -name "*.pftrace" -o -name "*agent_info.csv" \) -delete
# the wide-beam (configs[4]) task, 16 lanes: timing vs the reference's output + kernel trace
TASK_BEAM=1e-120 TASK_WBEAM=1e-80 tools/utt_task.sh wsj 16 400 "1 16" -pbeam 1e-100 -maxhmmpf 100000 > $OUT/wsj_task.txt 2>&1
TASK_BEAM=1e-120 TASK_WBEAM=1e-80 S3A_UTT=16 tools/prof_task.sh ${1:-r2}_wsj_utt16 wsj 16 400 -pbeam 1e-100 -maxhmmpf 100000 > $OUT/wsj_prof.txt 2>&1
# the hub4 task: one lane, and the frame-synchronous drop-in for comparison
tools/utt_task.sh hub4 64 600 "1 4 16 64" > $OUT/hub4_task.txt 2>&1
S3A_UTT=1 tools/prof_task.sh ${1:-r2}_hub4_utt1 hub4 4 600 > $OUT/hub4_utt1_prof.txt 2>&1
du -sh $OUT; tail -3 $OUT/pytest_gpu.log; tail -2 $OUT/smoke.log; tail -2 $OUT/bench.err; cat $OUT/wsj_task.txt | grep -v histogram; grep -v histogram $OUT/hub4_task.txt; cat $OUT/prof_summary.txt | head -60
