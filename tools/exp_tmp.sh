cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2z
timeout -k 10 1000 python -m pytest tests -m gpu -q > gpurun_out/r2z/gpu_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r2z/gpu_tests.log
tail -3 gpurun_out/r2z/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r2z/smoke.log 2>&1; tail -1 gpurun_out/r2z/smoke.log
timeout -k 10 600 python bench.py > gpurun_out/r2z/bench.json 2> gpurun_out/r2z/bench.err; echo "bench rc=$?"
