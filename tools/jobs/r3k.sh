#!/bin/bash
# Round 3, job k: the 4-pivot factor_subpanel (potrf tests, build/chol probe) and the hidden-layer arithmetic variants.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3k
mkdir -p $O
cd $R
python -m pytest tests/test_gpu_kernels.py -m gpu -q -x > $O/tests_kernels.log 2>&1
tail -3 $O/tests_kernels.log
python tools/build_chol_probe.py 1024 4096 16384 2>&1 | grep potrf | tee $O/build_chol.log
for v in i8 f64; do
echo "== hidden-layer A: $v (+ float64 C)"
NSGP_HIDDEN_A=$v python -m pytest tests/test_gpu_headline_precision.py -m gpu -q -s > $O/prec_$v.log 2>&1
grep -h "max-norm relative errors\|passed\|failed" $O/prec_$v.log
NSGP_HIDDEN_A=$v python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-build-chol > $O/bench_$v.json.log 2> $O/bench_$v.err
python -c "
import json
d=json.loads(open('$O/bench_$v.json.log').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['gemm_ms_per_step'], d.get('f64acc_projection'), d.get('i8_projection'))"
done
