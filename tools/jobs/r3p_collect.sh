#!/bin/bash
# Copies / summarises gpurun_out/r3p (tools/jobs/r3p_profiles.sh) into profiles/r03.  Run in the repo root, on the CPU box.
set -e
R=$(pwd)
O=$R/gpurun_out/r3p
if [ -z "$GRAFT_REPO_ROOT" ] && [ ! -d $O ]; then echo "no $O"; exit 1; fi
P=$R/profiles/r03
mkdir -p $P
for f in bench_default_run bench_profiled bench_rank_share_2 bench_rank_share_4 bench_rank_share_8 bench_cfg5_f32 bench_cfg5_bf16 bench_rehearse_rccl; do
  grep '^{' $O/$f.json.log | tail -1 > $P/$f.json.log
done
cp $(ls $O/kt/*/*kernel_stats.csv | tail -1) $P/bench_kernel_stats.csv
python tools/step_timeline.py $(ls $O/kt/*/*kernel_trace.csv | tail -1) > $P/step_timeline.txt
python tools/pmc_summary.py $O/pmc_fetch $O/pmc_write --match gemm_kernel > $P/gemm_hbm_pmc.txt
python tools/pmc_summary.py $O/pmc_fetch $O/pmc_write --match i8_ >> $P/gemm_hbm_pmc.txt
python tools/pmc_mfma_summary.py $O/pmc_sq --match gemm_kernel potrf i8_proj > $P/gemm_mfma_pmc.txt
cp $O/gemm_traffic.json $P/gemm_traffic.json
python tools/build_chol_profile_summary.py $O/bc_kt --write $O/bc_pmc_write --fetch $O/bc_pmc_fetch --sq $O/bc_pmc_sq --sizes 4096,16384 > $P/build_chol_profile.txt
cp $(ls $O/bc_kt/*/*kernel_stats.csv | tail -1) $P/build_chol_kernel_stats.csv
grep -v amdgpu $O/build_chol_probe_plain.log > $P/build_chol_probe.log
cp $(ls $O/kt_secondary/*/*kernel_stats.csv | tail -1) $P/secondary_steps_kernel_stats.csv
grep -v amdgpu $O/kt_secondary.log | tail -20 > $P/secondary_steps.log
for f in mfma_rate potrf_stamps gemm_stamps gemm_bench precision_after_training; do grep -v amdgpu $O/$f.log > $P/$f.log; done
grep -h "after .* Adam\|max-norm relative errors\|passed\|failed" $O/headline_precision.log > $P/headline_precision.log
ls -la $P
