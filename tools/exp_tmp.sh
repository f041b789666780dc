cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2y
timeout -k 10 600 python -m pytest tests/test_gpu_knn.py tests/test_gpu_fullsize.py::test_knn_1m_points_vs_oracle tests/test_gpu_laplacian.py tests/test_gpu_topology.py tests/test_gpu_radius.py -m gpu -q -x > gpurun_out/r2y/tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r2y/tests.log
tail -3 gpurun_out/r2y/tests.log
for i in 1 2; do
timeout -k 10 200 python bench.py --no-cpu --no-skeleton --no-ransac --no-rays --steps 5 > /tmp/b.json 2>/dev/null
python - <<PY
import json
d=json.loads(open("/tmp/b.json").read().strip().splitlines()[-1])
print("knn", round(d["knn"]["ms_per_step"],3), {k:round(v,3) for k,v in d["knn"]["phases_ms"].items()})
PY
done
