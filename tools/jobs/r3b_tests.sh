#!/bin/bash
# Round 3, job b: the GPU parity suite with the achieved errors printed (-s), the cfg5 bench line, the GEMM stamp probe.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3b
mkdir -p $O
cd $R
python -m pytest tests -m gpu -q -s > $O/gpu_tests.log 2>&1
rc=$?
grep -h "\[measured\]\|kappa\|max-norm\|configs\[" $O/gpu_tests.log > $O/measured.log
tail -5 $O/gpu_tests.log
python tools/probes/gemm_stamps.py > $O/gemm_stamps.log 2>&1
python bench.py --config cfg5 --steps 10 --warmup 3 > $O/bench_cfg5_f32.json.log 2> $O/bench_cfg5_f32.err
exit $rc
