#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3e
mkdir -p $O
cd $R
python tools/probes/gemm_stamps.py > $O/stamps_2wg.log 2>&1
NSGP_GEMM_LDS_EXTRA=40000 ONLY="C = Lq,Kzxbar,dense,Wbar" python tools/probes/gemm_stamps.py > $O/stamps_1wg.log 2>&1
tail -3 $O/stamps_1wg.log
