cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3a
timeout -k 10 600 python -m pytest tests/test_gpu_dbscan.py tests/test_gpu_golden_dbscan.py tests/test_gpu_fullsize.py::test_dbscan_1m_points_vs_oracle tests/test_gpu_config5.py::test_dbscan_5m_points_vs_oracle tests/test_gpu_wrappers.py tests/test_gpu_threads.py -m gpu -q -x > gpurun_out/r3a/tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r3a/tests.log
tail -3 gpurun_out/r3a/tests.log
for i in 1 2 3; do
timeout -k 10 200 python bench.py --no-cpu --no-skeleton --no-ransac --no-knn --no-rays --steps 50 > /tmp/b.json 2>/dev/null
python - <<PY
import json
d=json.loads(open("/tmp/b.json").read().strip().splitlines()[-1])
print("dbscan", round(d["ms_per_step"],4), round(d["value"],1), round(d["roofline"]["frac"],3), {k:round(v["avg_ms"],4) for k,v in d["kernels"].items()})
PY
done
