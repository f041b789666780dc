cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2p
timeout -k 10 600 python -m pytest tests/test_gpu_lbc.py tests/test_gpu_laplacian.py tests/test_gpu_config3.py::test_every_solve_of_a_loop_within_1e5 tests/test_gpu_config3.py::test_gpu_loop_solves_against_superlu -m gpu -q > gpurun_out/r2p/tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r2p/tests.log
tail -3 gpurun_out/r2p/tests.log
for v in 1 0 1 0; do
PYQSM_AMG_AP=$v timeout -k 10 120 python examples/config3_skeleton.py --points 1000000 --contraction 3 > gpurun_out/r2p/c3_ap$v.json 2>&1
tail -1 gpurun_out/r2p/c3_ap$v.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('AP=$v 1M', round(d['wall_s'],3), d['solve_outer_iterations'], d['solve_multigrid_cg_iterations'], round(d['solve_multigrid_cg_ms']/d['solve_multigrid_cg_iterations'],4), round(d['solve_multigrid_setup_ms'],1))"
PYQSM_AMG_AP=$v timeout -k 10 120 python examples/config3_skeleton.py --points 50000 --contraction 3 > gpurun_out/r2p/c3_50k_ap$v.json 2>&1
tail -1 gpurun_out/r2p/c3_50k_ap$v.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('AP=$v 50k', round(d['wall_s'],3), d['solve_outer_iterations'], d['solve_multigrid_cg_iterations'], round(d['solve_multigrid_cg_ms']/d['solve_multigrid_cg_iterations'],4), round(d['solve_multigrid_setup_ms'],1))"
done
