#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3d
mkdir -p $O
cd $R
python -m pytest tests/test_gpu_svgp.py tests/test_gpu_headline_precision.py tests/test_gpu_dgp.py tests/test_gpu_goldens.py tests/test_gpu_bf16.py -m gpu -q -s > $O/tests.log 2>&1
tail -6 $O/tests.log
grep -h "after .* Adam\|max-norm relative errors\|f64-Kzx" $O/tests.log
python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-build-chol > $O/bench.json.log 2> $O/bench.err
python -c "
import json
d=json.loads(open('$O/bench.json.log').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['f64acc_projection'])"
