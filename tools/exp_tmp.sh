cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2m
timeout -k 10 600 python -m pytest tests/test_gpu_batch.py tests/test_gpu_laplacian.py -m gpu -q -s > gpurun_out/r2m/tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r2m/tests.log
grep "first step\|passed\|failed" gpurun_out/r2m/tests.log
timeout -k 10 900 python examples/config5_pipeline.py --scale 1 --skeleton-iters 20 --max-trees 100 --workers 8 > gpurun_out/r2m/config5_batch.json 2> gpurun_out/r2m/config5_batch.err
tail -1 gpurun_out/r2m/config5_batch.json | cut -c1-600
