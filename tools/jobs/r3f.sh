#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3f
mkdir -p $O
cd $R
python tools/tmp/dbg_gemm.py 2>&1 | grep rep | head -6
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_svgp.py tests/test_gpu_dgp.py tests/test_gpu_dist.py tests/test_gpu_goldens.py -m gpu -q > $O/tests.log 2>&1
tail -5 $O/tests.log
python tools/gemm_bench.py > $O/gemm_bench_last.log 2>&1
NCOLS=4096 BATCH=2 python tools/gemm_bench.py > $O/gemm_bench_hidden.log 2>&1
cat $O/gemm_bench_last.log $O/gemm_bench_hidden.log
ONLY="C = Lq,Kzxbar,Wbar" python tools/probes/gemm_stamps.py > $O/stamps_2wg.log 2>&1
python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-build-chol > $O/bench.json.log 2> $O/bench.err
python -c "
import json
d=json.loads(open('$O/bench.json.log').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['gemm_ms_per_step'], d['f64acc_projection']['ms_per_step'])"
