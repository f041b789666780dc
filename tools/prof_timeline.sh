#!/bin/bash
# kernel timeline of a script: bash tools/prof_timeline.sh <name> <first_kernel> <count> <script.py> [args]
name=$1; first=$2; count=$3; shift 3
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=/tmp/tl_$name
rm -rf $O; mkdir -p $O
timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d $O -o p -- python3 $R/"$@" > $O/out.log 2> $O/err.log
cd $R
f=$(find $O -name "*kernel_trace.csv" | head -1)
python3 tools/kernel_timeline.py "$f" "$first" "$count" > gpurun_out/${name}_timeline.txt 2>&1
tail -2 $O/out.log
