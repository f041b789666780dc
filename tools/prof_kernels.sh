#!/bin/bash
# per-kernel times of one python script under rocprofv3 (run through gpurun from the repo root):
#   bash tools/prof_kernels.sh <name> <script.py> [args...]   -> gpurun_out/<name>_kernel_stats.csv
name=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=/tmp/prof_$name
rm -rf $O; mkdir -p $O
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o p -- python3 $R/"$@" > $O/out.log 2> $O/err.log
cd $R
f=$(find $O -name "*kernel_stats.csv" | head -1)
mkdir -p gpurun_out
cp "$f" gpurun_out/${name}_kernel_stats.csv 2>/dev/null
cp $O/out.log gpurun_out/${name}_out.log
grep -v rocprofv3 $O/err.log | tail -5
