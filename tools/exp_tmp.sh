cd $GRAFT_REPO_ROOT
for w in 0 0.6 0.7 0.8 0.9; do
PYQSM_AMG_OMEGA=$w timeout -k 10 120 python examples/config3_skeleton.py --points 1000000 --contraction 3 > /tmp/c3.json 2>&1
tail -1 /tmp/c3.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('omega=$w 1M', round(d['wall_s'],3), d['solve_outer_iterations'], d['solve_multigrid_cg_iterations'], round(d['solve_outer_ms_incl_inner']))"
done
