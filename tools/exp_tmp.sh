cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2u
timeout -k 10 1000 python -m pytest tests -m gpu -q --durations=5 > gpurun_out/r2u/gpu_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r2u/gpu_tests.log
grep "passed\|failed\|FAILED" gpurun_out/r2u/gpu_tests.log
timeout -k 10 600 python bench.py > gpurun_out/r2u/bench.json 2> gpurun_out/r2u/bench.err; echo "bench rc=$?"
python - <<PY
import json
d=json.loads(open("gpurun_out/r2u/bench.json").read().strip().splitlines()[-1])
print(round(d["value"],1), round(d["ms_per_step"],4), round(d["roofline"]["frac"],3), round(d["knn"]["ms_per_step"],3), {k:(round(v["wall_s"],2), v["solves_not_converged"]) for k,v in d["skeleton"]["rows"].items()})
PY
