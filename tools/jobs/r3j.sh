#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3j
mkdir -p $O
cd $R
python -m pytest tests/test_gpu_i8.py tests/test_gpu_headline_precision.py tests/test_gpu_svgp.py tests/test_gpu_dgp.py tests/test_gpu_goldens.py tests/test_gpu_bf16.py tests/test_gpu_dist.py -m gpu -q -s > $O/tests.log 2>&1
tail -4 $O/tests.log
grep -h "after .* Adam\|max-norm relative errors" $O/tests.log
python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-build-chol > $O/bench.json.log 2> $O/bench.err
python -c "
import json
d=json.loads(open('$O/bench.json.log').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['gemm_ms_per_step'], d.get('f64acc_projection'), d.get('i8_projection'))"
python bench.py --config cfg5 --steps 10 --warmup 3 --no-cpu-baseline > $O/cfg5_f32.json.log 2>/dev/null
python bench.py --config cfg5 --forward bf16 --steps 10 --warmup 3 --no-cpu-baseline > $O/cfg5_bf16.json.log 2>/dev/null
python -c "
import json
for f in ('cfg5_f32','cfg5_bf16'):
    d=json.loads(open('$O/'+f+'.json.log').read().strip().splitlines()[-1])
    print(f, d['value'], d['ms_per_step'], d['roofline']['frac'], d.get('i8_projection'), d.get('bf16_projection'))"
