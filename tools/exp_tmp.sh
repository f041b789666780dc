cd $GRAFT_REPO_ROOT
for v in 3 2 1.5 1 4; do
PYQSM_KNN_OCC=$v timeout -k 10 200 python bench.py --no-cpu --no-skeleton --no-ransac --no-rays --steps 5 > /tmp/b.json 2>/dev/null
python - <<PY
import json
d=json.loads(open("/tmp/b.json").read().strip().splitlines()[-1])
print("OCC=k/$v", round(d["knn"]["ms_per_step"],3), {k:round(v,3) for k,v in d["knn"]["phases_ms"].items()})
PY
done
