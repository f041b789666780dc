cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2k
for cfg in "1 1" "1 8" "1 16" "4 2" "4 4" "6 3"; do set -- $cfg
timeout -k 10 300 python tools/exp_trees.py --trees 24 --procs $1 --threads $2 >> gpurun_out/r2k/trees.log 2>&1
done
cat gpurun_out/r2k/trees.log | grep trees=
