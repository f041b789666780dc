cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2r
echo AP=1; PYQSM_AMG_AP=1 timeout -k 10 300 python tools/exp_conv_tmp.py 4 2>&1 | grep "^{"
echo AP=0; PYQSM_AMG_AP=0 timeout -k 10 300 python tools/exp_conv_tmp.py 4 2>&1 | grep "^{"
