#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3l
mkdir -p $O
cd $R
python -m pytest tests/test_gpu_kernels.py -m gpu -q -x > $O/tests_kernels.log 2>&1
tail -3 $O/tests_kernels.log
rm -f $O/potrf_stamps.log
for dt in float32 float64; do
python tools/probes/potrf_stamps.py 1024 $dt 2>&1 | grep -v amdgpu | tee -a $O/potrf_stamps.log
done
python tools/build_chol_probe.py 1024 4096 16384 2>&1 | grep potrf | tee $O/build_chol.log
python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-build-chol > $O/bench.json.log 2> $O/bench.err
python -c "
import json
d=json.loads(open('$O/bench.json.log').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['gemm_ms_per_step'])"
