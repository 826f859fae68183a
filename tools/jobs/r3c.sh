#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3c
mkdir -p $O
cd $R
python tools/probes/precision_after_training.py 25 1000 > $O/precision.log 2>&1
python -m pytest tests/test_gpu_cfg2.py tests/test_gpu_dist.py -m gpu -q -s -k "M512 or capture_failure or distinct" > $O/tests.log 2>&1
tail -3 $O/tests.log
