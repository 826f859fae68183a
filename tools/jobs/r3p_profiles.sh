#!/bin/bash
# Round 3: the committed evidence under profiles/r03 (run on the GPU box: gpurun -- bash tools/jobs/r3p_profiles.sh;
# tools/jobs/r3p_collect.sh then copies the summaries into profiles/r03)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3p
rm -rf $O
mkdir -p $O
cd $R
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu-baseline --no-build-chol"
# PMC passes first: roofline.traffic of the bench lines below comes from the hash-gated JSON they produce
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- $B --steps 3 --warmup 1 --no-graph > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- $B --steps 3 --warmup 1 --no-graph > $O/pmc_write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT \
    --kernel-trace --output-format csv -d $O/pmc_sq -- $B --steps 3 --warmup 1 --no-graph > $O/pmc_sq.log 2>&1
cd $R
python tools/gemm_traffic.py $O/pmc_fetch $O/pmc_write > $O/gemm_traffic.json
cp $O/gemm_traffic.json $R/profiles/r03/gemm_traffic.json
python bench.py > $O/bench_default_run.json.log 2> $O/bench_default_run.err
echo "bench done"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- $B --steps 30 --warmup 5 > $O/bench_profiled.json.log 2>&1
echo "bench profiles done"
# second half of the metric: Gibbs build + potrf at N = 4096 / 16384
P="python3 $R/tools/build_chol_probe.py 4096 16384"
$P > $O/build_chol_probe_plain.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bc_kt -- $P > $O/bc_kt.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/bc_pmc_write -- $P > $O/bc_pmc_write.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/bc_pmc_fetch -- $P > $O/bc_pmc_fetch.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT \
    --kernel-trace --output-format csv -d $O/bc_pmc_sq -- $P > $O/bc_pmc_sq.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_secondary -- python3 $R/tools/secondary_steps_probe.py > $O/kt_secondary.log 2>&1
echo "build/chol profiles done"
cd $R
for g in 2 4 8; do python bench.py --no-cpu-baseline --no-build-chol --steps 100 --warmup 10 --rank-share $g > $O/bench_rank_share_$g.json.log 2>/dev/null; done
python bench.py --config cfg5 --steps 10 --warmup 3 > $O/bench_cfg5_f32.json.log 2>/dev/null
python bench.py --config cfg5 --forward bf16 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_cfg5_bf16.json.log 2>/dev/null
python bench.py --no-cpu-baseline --no-build-chol --steps 100 --warmup 10 --rehearse-rccl > $O/bench_rehearse_rccl.json.log 2>/dev/null
echo "bench variants done"
python tools/probes/mfma_rate.py > $O/mfma_rate.log 2>&1
for dt in float32 float64; do python tools/probes/potrf_stamps.py 1024 $dt >> $O/potrf_stamps.log 2>&1; done
python tools/probes/gemm_stamps.py > $O/gemm_stamps.log 2>&1
python tools/gemm_bench.py > $O/gemm_bench.log 2>&1
python -m pytest tests/test_gpu_headline_precision.py -m gpu -q -s > $O/headline_precision.log 2>&1
python tools/probes/precision_after_training.py 25 1000 > $O/precision_after_training.log 2>&1
echo done
