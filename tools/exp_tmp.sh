cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python -m pytest tests -m gpu -q -x > gpurun_out/gpu_tests.log 2>&1
rc=$?
tail -5 gpurun_out/gpu_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
