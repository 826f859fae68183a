#!/usr/bin/env python3
"""MFMA-pipe utilisation, wait fractions and LDS bank conflicts per kernel from a rocprofv3 --pmc pass
(SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT):
    python tools/pmc_mfma_summary.py <pmc output dir> [--match gemm_kernel potrf_step]"""
import collections
import csv
import glob
import sys

d = sys.argv[1]
match = sys.argv[sys.argv.index('--match') + 1:] if '--match' in sys.argv else ['gemm_kernel', 'potrf_step', 'bf16']
f = glob.glob(d + '/**/*counter_collection.csv', recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    n = r['Kernel_Name']
    if not any(m in n for m in match):
        continue
    key = n[n.index('::') + 2 if '::' in n else 0:][:75] + '  grid=' + r['Grid_Size']
    acc[key][r['Counter_Name']].append(float(r['Counter_Value']))
print('# mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES); the other columns are fractions of SQ_WAVE_CYCLES')
for k, c in sorted(acc.items()):
    m = {n: sum(v) / len(v) for n, v in c.items()}
    wc = m['SQ_WAVE_CYCLES']
    print('%-95s n=%3d mfma_util=%.3f wait_any=%.2f issue_stall=%.2f active=%.2f lds_conflict=%.3f' % (
        k, len(c['SQ_WAVE_CYCLES']), m['SQ_VALU_MFMA_BUSY_CYCLES'] / (4 * m['SQ_BUSY_CU_CYCLES']), m['SQ_WAIT_ANY'] / wc,
        m['SQ_WAIT_INST_ANY'] / wc, m['SQ_ACTIVE_INST_ANY'] / wc, m['SQ_LDS_BANK_CONFLICT'] / wc))
