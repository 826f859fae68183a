#!/usr/bin/env python3
"""L2<->fabric bytes per launch of the DSVI step's f32 GEMM family from two rocprofv3 PMC passes
(`--pmc FETCH_SIZE` and `--pmc WRITE_SIZE`, each with --kernel-trace only) of
`bench.py --steps 3 --warmup 1 --no-graph --no-cpu-baseline --no-build-chol` (every f32 128-row-tile GEMM dispatch of
such a run belongs to a DSVI step).

    python tools/gemm_traffic.py <fetch-pass-dir> <write-pass-dir> > profiles/r03/gemm_traffic.json

FETCH_SIZE / WRITE_SIZE are in KiB; FETCH_SIZE is doubled (gfx950 counts 64 B per 128-B request on wide streaming
reads, MI355X_MICROARCH.md) and includes Infinity-Cache hits, so the figure is an upper bound on HBM traffic.
"""
import csv
import glob
import json
import os
import sys

def total(d, counter):
    tot, n = 0.0, 0
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            if 'gemm_kernel<float, 128' in r['Kernel_Name'] and r['Counter_Name'] == counter:
                tot += float(r['Counter_Value']) * 1024.0
                n += 1
    return tot, n


def gemm_source_sha():
    import hashlib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    return hashlib.sha256(open(os.path.join(root, 'nonstationary-precip_amd', 'csrc', 'gemm.hip'), 'rb').read()).hexdigest()[:16]


def main():
    fetch, nf = total(sys.argv[1], 'FETCH_SIZE')
    write, nw = total(sys.argv[2], 'WRITE_SIZE')
    # algorithmic operand + result bytes of the step's f32 GEMM launches (headline config: M = 1024; hidden layer b = 2,
    # n = 4096; last layer b = 1, n = 40960): per layer 3 products of an M x M operand with an M x n one into M x n
    # (C = Lq^T A, Abar = Lq C [+ A read in its epilogue], Kzxbar = W^T Abar) and 2 long-K products of two M x n operands
    # into M x M (Lqbar, Wbar); A = W Kzx runs in float64 arithmetic (a different kernel, not counted here)
    # (round 3: the hidden layer's C = Lq^T A also runs on the float64-accumulating kernel, settings.hidden_kzx_f64, so the
    # hidden layer has 2 single-pass products here and the last layer 3: 9 launches per step)
    M = 1024
    alg, launches = 0, 0
    for b, n, single in ((2, 4096, 2), (1, 40960, 3)):
        mn, mm = 4 * b * M * n, 4 * b * M * M
        alg += single * (mm // 2 + 2 * mn) + mn + 2 * (2 * mn + mm // 2)
        launches += single + 2
    out = {'kernel': 'gemm_kernel<float,128,128|64,...> launches of a DSVI step', 'launches_counted': nf,
           'gemm_source_sha': gemm_source_sha(), 'algorithmic_bytes_per_launch': alg / float(launches),
           'f32_gemm_launches_per_step': launches,
           'fetch_bytes_per_launch_x2': 2.0 * fetch / max(nf, 1), 'write_bytes_per_launch': write / max(nw, 1),
           'bytes_per_launch': 2.0 * fetch / max(nf, 1) + write / max(nw, 1),
           'source': 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only); '
                     'FETCH doubled per the gfx950 correction; counts Infinity-Cache hits'}
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == '__main__':
    main()
