#!/bin/bash
# Round-3 profiles on the GPU box (run through gpurun from the repo root):
#   bash tools/profile_r03.sh
# Raw rocprofv3 output stays in /tmp on the box; tools/profile_summary_r03.py condenses it into
# gpurun_out/prof_r03_summary/r03_* (copy those into profiles/). The program goes directly after
# `--` (no env / bash hop under rocprofv3); PMC passes carry only --kernel-trace; FETCH_SIZE and
# WRITE_SIZE in separate passes (TCC slots). hipGraph replays stay ON under the profiler.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=/tmp/prof_r03
rm -rf $O; mkdir -p $O
DB="--steps 5 --warmup 1 --no-rays --no-knn --no-skeleton --no-ransac --no-cpu"
KN="--steps 2 --warmup 1 --no-rays --no-skeleton --no-ransac --no-cpu"
# a pass that runs into its limit ends the script: no further GPU step after a kill
run() { name=$1; shift; timeout -k 10 500 rocprofv3 "$@" > $O/$name.json 2> $O/$name.err; rc=$?; echo "$name rc=$rc" >> $O/steps.log;
        if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then mkdir -p $R/gpurun_out/prof_r03_summary; cp $O/steps.log $R/gpurun_out/prof_r03_summary/; echo "$name timed out: stopping"; exit 1; fi; }
run bench      --kernel-trace --stats --output-format csv -d $O/bench -o bench -- python3 $R/bench.py --no-cpu --no-config5
run fetch      --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -o f -- python3 $R/bench.py $DB
run write      --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -o w -- python3 $R/bench.py $DB
run sq_dbscan  --kernel-trace --pmc SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES SQ_WAVE_CYCLES --output-format csv -d $O/sq_dbscan -o s -- python3 $R/bench.py $DB
run sq_knn     --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES SQ_ACTIVE_INST_ANY --output-format csv -d $O/sq_knn -o s -- python3 $R/bench.py $KN
run knn_fetch  --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/knn_fetch -o f -- python3 $R/bench.py $KN
run knn_write  --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/knn_write -o w -- python3 $R/bench.py $KN
run lap_fetch  --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/lap_fetch -o f -- python3 $R/tools/prof_skel.py lap
run lap_write  --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/lap_write -o w -- python3 $R/tools/prof_skel.py lap
run lap_stats  --kernel-trace --stats --output-format csv -d $O/lap_stats -o s -- python3 $R/tools/prof_skel.py lap
run sol_fetch  --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/sol_fetch -o f -- python3 $R/tools/prof_skel.py solve
run sol_write  --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/sol_write -o w -- python3 $R/tools/prof_skel.py solve
run sol_stats  --kernel-trace --stats --output-format csv -d $O/sol_stats -o s -- python3 $R/tools/prof_skel.py solve
# both hipGraph paths of the solver under the profiler (the amg line was missing)
timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/graph_jacobi -o g -- python3 $R/tools/graph_under_rocprof.py jacobi > $O/graph_jacobi.log 2>&1
export PYQSM_AMG_GRAPH=1
timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/graph_amg -o g -- python3 $R/tools/graph_under_rocprof.py amg > $O/graph_amg.log 2>&1
unset PYQSM_AMG_GRAPH
cd $R
mkdir -p gpurun_out/prof_r03_summary
python3 tools/profile_summary_r03.py $O gpurun_out/prof_r03_summary > gpurun_out/prof_r03_summary/summary.log 2>&1
cp $O/steps.log gpurun_out/prof_r03_summary/
for f in $O/*.err; do echo "== $f"; grep -v "rocprofv3\]\|Opened result" $f | tail -3; done > gpurun_out/prof_r03_summary/stderr_tails.log 2>&1
tail -5 gpurun_out/prof_r03_summary/summary.log
