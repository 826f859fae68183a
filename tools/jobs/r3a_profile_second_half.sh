#!/bin/bash
# Round 3, job a: rocprofv3 evidence for the metric's second half (Gibbs K build + potrf) and the secondary steps.
# Run on the GPU box:  gpurun -- bash tools/jobs/r3a_profile_second_half.sh
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3a
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
P="python3 $R/tools/build_chol_probe.py 4096 16384"
$P > $O/probe_plain.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- $P > $O/kt.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- $P > $O/pmc_write.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- $P > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT \
    --kernel-trace --output-format csv -d $O/pmc_sq -- $P > $O/pmc_sq.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_secondary -- python3 $R/tools/secondary_steps_probe.py > $O/kt_secondary.log 2>&1
echo done
