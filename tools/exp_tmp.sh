set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2e
for m in 0 4 6 8; do
  PYQSM_AMG_FIXED=$m timeout -k 10 120 python examples/config3_skeleton.py --points 1000000 --contraction 3 > gpurun_out/r2e/c3_1m_fixed$m.json 2>&1
  PYQSM_AMG_FIXED=$m timeout -k 10 120 python examples/config3_skeleton.py --points 50000 --contraction 3 > gpurun_out/r2e/c3_50k_fixed$m.json 2>&1
done
PYQSM_AMG_FIXED=4 timeout -k 10 300 python tools/solver_accuracy.py --points 20000 --contraction 7 > gpurun_out/r2e/acc_c7_fixed4.log 2>&1
PYQSM_AMG_FIXED=6 timeout -k 10 300 python tools/solver_accuracy.py --points 20000 --contraction 7 > gpurun_out/r2e/acc_c7_fixed6.log 2>&1
timeout -k 10 600 python -m pytest tests/test_gpu_config3.py tests/test_gpu_rays_f64.py -m gpu -q -s > gpurun_out/r2e/tests.log 2>&1
echo finished
