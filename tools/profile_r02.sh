#!/bin/bash
# Round-2 profiles on the GPU box (run through gpurun from the repo root):
#   bash tools/profile_r02.sh
# Raw rocprofv3 output goes to /tmp on the box; tools/profile_summary.py condenses it into
# gpurun_out/prof_r02_summary/r02_* (copy those into profiles/). The program goes directly after `--` (no env/bash hop under rocprofv3). PMC passes
# carry only --kernel-trace. hipGraph replays stay ON under the profiler (the round-1 workaround that
# sniffed ROCP_TOOL_LIBRARIES is gone; tools/graph_under_rocprof.py is the regression check).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=/tmp/prof_r02          # raw traces are large: they stay on the box
rm -rf $O; mkdir -p $O
DB="--steps 3 --warmup 1 --no-rays --no-knn --no-skeleton --no-ransac --no-cpu"
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -o bench -- python3 $R/bench.py --no-cpu > $O/bench.json 2> $O/bench.err
timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -o f -- python3 $R/bench.py $DB > $O/fetch.json 2> $O/fetch.err
timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -o w -- python3 $R/bench.py $DB > $O/write.json 2> $O/write.err
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES SQ_WAVE_CYCLES --output-format csv -d $O/sq_dbscan -o s -- python3 $R/bench.py $DB > $O/sq_dbscan.json 2> $O/sq_dbscan.err
# the brute-force ray sweep: is the "10 executed flop, 73 % of the issue bound" claim counter-backed?
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_WAVES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/sq_rays -o s -- python3 $R/bench.py --steps 2 --warmup 1 --no-knn --no-skeleton --no-ransac --no-cpu --ray-steps 1 > $O/sq_rays.json 2> $O/sq_rays.err
# kNN search kernel
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES SQ_ACTIVE_INST_ANY --output-format csv -d $O/sq_knn -o s -- python3 $R/bench.py --steps 2 --warmup 1 --no-rays --no-skeleton --no-ransac --no-cpu > $O/sq_knn.json 2> $O/sq_knn.err
# both hipGraph paths of the solver under the profiler
timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/graph_jacobi -o g -- python3 $R/tools/graph_under_rocprof.py jacobi > $O/graph_jacobi.log 2>&1
cd $R
mkdir -p gpurun_out/prof_r02_summary
python3 tools/profile_summary.py $O gpurun_out/prof_r02_summary > gpurun_out/prof_r02_summary/summary.log 2>&1
for f in $O/*.err; do echo "== $f"; tail -3 $f; done > gpurun_out/prof_r02_summary/stderr_tails.log 2>&1
tail -5 gpurun_out/prof_r02_summary/summary.log
