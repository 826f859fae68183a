#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3m
mkdir -p $O
cd $R
for o in 0 1 2 3; do
NSGP_I8_DBG=$o python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-build-chol > $O/bench_d$o.json.log 2> $O/bench_d$o.err
python -c "
import json
d=json.loads(open('$O/bench_d$o.json.log').read().strip().splitlines()[-1])
print('dbg $o', d['value'], d['ms_per_step'], d.get('i8_projection')['ms_per_step'])"
done
