#!/bin/bash
# Round 3: the committed evidence under profiles/r03 (run on the GPU box: gpurun -- bash tools/jobs/r3p_profiles.sh)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3p
mkdir -p $O
cd $R
python bench.py > $O/bench_default_run.json.log 2> $O/bench_default_run.err
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu-baseline --no-build-chol"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- $B --steps 30 --warmup 5 > $O/bench_profiled.json.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- $B --steps 3 --warmup 1 --no-graph > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- $B --steps 3 --warmup 1 --no-graph > $O/pmc_write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT \
    --kernel-trace --output-format csv -d $O/pmc_sq -- $B --steps 3 --warmup 1 --no-graph > $O/pmc_sq.log 2>&1
cd $R
for g in 2 4 8; do python bench.py --no-cpu-baseline --no-build-chol --steps 100 --warmup 10 --rank-share $g > $O/bench_rank_share_$g.json.log 2>/dev/null; done
python bench.py --config cfg5 --steps 10 --warmup 3 > $O/bench_cfg5_f32.json.log 2>/dev/null
python bench.py --config cfg5 --forward bf16 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_cfg5_bf16.json.log 2>/dev/null
python bench.py --no-cpu-baseline --no-build-chol --steps 100 --warmup 10 --rehearse-rccl > $O/bench_rehearse_rccl.json.log 2>/dev/null
echo done
