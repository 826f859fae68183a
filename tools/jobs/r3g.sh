#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3g
mkdir -p $O
cd $R
python -m pytest tests -m gpu -q -x > $O/tests.log 2>&1
tail -4 $O/tests.log
python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-build-chol > $O/bench.json.log 2> $O/bench.err
python -c "
import json
d=json.loads(open('$O/bench.json.log').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['gemm_ms_per_step'], d['f64acc_projection'])"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-build-chol > $O/bench_profiled.json.log 2>&1
python3 $R/tools/step_timeline.py $O/kt > $O/step_timeline.txt 2>&1
tail -30 $O/step_timeline.txt
