cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2o
timeout -k 10 1000 python -m pytest tests -m gpu -q --durations=8 > gpurun_out/r2o/gpu_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r2o/gpu_tests.log
grep "per-step\|worst\|first step\|passed\|failed\|FAILED" gpurun_out/r2o/gpu_tests.log
