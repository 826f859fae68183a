#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3h
mkdir -p $O
cd $R
python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "potrf or trtri or chol" > $O/tests.log 2>&1
tail -3 $O/tests.log
for nb2 in 16 32; do
for la in 0 1; do
echo "== NB2=$nb2 LOOKAHEAD=$la"
NSGP_POTRF_NB2=$nb2 NSGP_POTRF_LOOKAHEAD=$la python tools/build_chol_probe.py 4096 8192 16384 2>&1 | grep potrf
done
done
