#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3n
rm -rf $O; mkdir -p $O
cd $R
python -m pytest tests/test_gpu_svgp.py tests/test_gpu_dgp.py tests/test_gpu_kernels.py -m gpu -q -x > $O/tests.log 2>&1; tail -2 $O/tests.log
python tools/probes/gemm_stamps.py > $O/gemm_stamps.log 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/bench.py --no-cpu-baseline --no-build-chol --steps 30 --warmup 5 > $O/bench_profiled.json.log 2>&1
grep rowdot $O/kt/*/*kernel_stats.csv | cut -c1-200
