cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2t
timeout -k 10 600 python -m pytest tests/test_gpu_native_loop.py -m gpu -q -s > gpurun_out/r2t/tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r2t/tests.log
grep "first step\|passed\|failed\|Error\|^E  " gpurun_out/r2t/tests.log | head -20
timeout -k 10 300 python tools/dbg_native_tmp.py 2>&1 | grep "bounds\|keep_steps"
for e in python native; do
timeout -k 10 120 python examples/config3_skeleton.py --points 1000000 --contraction 3 --engine $e > gpurun_out/r2t/c3_$e.json 2>&1
tail -1 gpurun_out/r2t/c3_$e.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$e 1M', round(d['wall_s'],3), d['solve_outer_iterations'], d['solve_multigrid_cg_iterations'], round(d['laplacian_ms']), round(d['solve_outer_ms_incl_inner']))"
done
