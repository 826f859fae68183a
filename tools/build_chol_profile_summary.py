#!/usr/bin/env python3
"""Summary of rocprofv3 passes over tools/build_chol_probe.py (Gibbs K build + potrf, the metric's second half).

    python tools/build_chol_profile_summary.py <kernel-trace dir> [--write DIR] [--fetch DIR] [--sq DIR] [--sizes 4096,16384]

The probe runs, per (N, dtype), four repetitions of [gibbs build, potrf]: every Gibbs `pairwise_fwd_kernel` dispatch
opens a segment and the dispatches up to the next build are that repetition's factorisation.  Reported per (N, dtype),
fastest repetition: build duration and algorithmic GB/s (s (N^2 + 8 N) bytes), potrf device time (sum of its launches,
and the span from first start to last end) and TFLOP/s (N^3/3 over the span), launches, time by potrf kernel family.
PMC passes (each its own run, same dispatch order): WRITE_SIZE / FETCH_SIZE of the build kernel (FETCH doubled, the
gfx950 correction of MI355X_MICROARCH.md), and the MFMA-pipe utilisation of the factorisation's kernels
(SQ_VALU_MFMA_BUSY_CYCLES / (4 SQ_BUSY_CU_CYCLES), summed over the launches of one factorisation)."""
import collections
import csv
import glob
import re
import sys

REPS = 4          # tools/build_chol_probe.py: reps + 1 repetitions per (N, dtype)


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'^void ', '', name)
    return re.sub(r'\(.*$', '', name)


def is_build(name):
    return 'pairwise_fwd_kernel' in name and 'GibbsOp' in name


def is_chol(name):
    return any(s in name for s in ('potrf', 'gemm_kernel', 'trtri'))


def dur(r):
    return (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3


def load_trace(d):
    f = glob.glob(d + '/**/*kernel_trace.csv', recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    return rows


def load_pmc(d):
    f = glob.glob(d + '/**/*counter_collection.csv', recursive=True)[0]
    out = {}
    for r in csv.DictReader(open(f)):
        e = out.setdefault(int(r['Dispatch_Id']), {'name': r['Kernel_Name']})
        e[r['Counter_Name']] = float(r['Counter_Value'])
    return [out[k] for k in sorted(out)]


def blocks(seq, name_of):
    """[(build, [factorisation dispatches])] in dispatch order."""
    out = []
    for r in seq:
        n = name_of(r)
        if is_build(n):
            out.append((r, []))
        elif out and is_chol(n):
            out[-1][1].append(r)
    return out


def main():
    a = sys.argv[1:]
    opt = {}
    for k in ('--write', '--fetch', '--sq', '--sizes'):
        if k in a:
            i = a.index(k)
            opt[k] = a[i + 1]
            del a[i:i + 2]
    sizes = [int(x) for x in opt.get('--sizes', '4096,16384').split(',')]

    def label(i):                         # block index -> (N, dtype): sizes outer, (f32, f64) inner, REPS repetitions each
        j = i // REPS
        return sizes[j // 2] if j // 2 < len(sizes) else 0, ('f32', 'f64')[j % 2]

    segs = blocks(load_trace(a[0]), lambda r: r['Kernel_Name'])
    print('# kernel trace: fastest repetition per (N, dtype)')
    print('%-7s %-4s %10s %9s %9s | %10s %10s %8s %7s' % ('N', 'dt', 'build_us', 'GB/s', 'of 8 TB/s', 'potrf_us', 'span_us',
                                                          'TFLOP/s', 'launch'))
    fams = {}
    for i in range(0, len(segs), REPS):
        N, dt = label(i)
        reps = segs[i:i + REPS]
        s = 4 if dt == 'f32' else 8
        bt = min(dur(b) for b, _ in reps)
        best = min(reps[1:] or reps, key=lambda bp: sum(dur(r) for r in bp[1]))
        pt = sum(dur(r) for r in best[1])
        span = (max(int(r['End_Timestamp']) for r in best[1]) - min(int(r['Start_Timestamp']) for r in best[1])) / 1e3
        gbs = s * (N * N + 8 * N) / bt / 1e3
        print('%-7d %-4s %10.1f %9.1f %9.3f | %10.1f %10.1f %8.2f %7d' % (N, dt, bt, gbs, gbs / 8000, pt, span,
                                                                         N ** 3 / 3 / span / 1e6, len(best[1])))
        fam = collections.defaultdict(lambda: [0, 0.0])
        for r in best[1]:
            k = short(r['Kernel_Name'])[:60]
            fam[k][0] += 1
            fam[k][1] += dur(r)
        fams[(N, dt)] = fam
    print('\n# potrf time by kernel (fastest repetition)')
    for (N, dt), fam in fams.items():
        for k, (c, t) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
            print('%-7d %-4s %5d x %-62s %10.1f us' % (N, dt, c, k, t))

    for k, ctr in (('--write', 'WRITE_SIZE'), ('--fetch', 'FETCH_SIZE')):
        if k not in opt:
            continue
        print(f'\n# {ctr} of the build kernel (rocprofv3 reports KiB; mean over the repetitions)')
        segs_p = blocks(load_pmc(opt[k]), lambda r: r['name'])
        for i in range(0, len(segs_p), REPS):
            N, dt = label(i)
            s = 4 if dt == 'f32' else 8
            v = [b[ctr] for b, _ in segs_p[i:i + REPS]]
            m = sum(v) / len(v) * 1024 / 1e6
            extra = f' (x2 = {2 * m:.1f} MB: gfx950 correction)' if ctr == 'FETCH_SIZE' else ''
            print('%-7d %-4s %s = %.1f MB%s; algorithmic: %.1f MB written, %.2f MB read' % (
                N, dt, ctr, m, extra, s * N * N / 1e6, s * 8 * N / 1e6))
    if '--sq' in opt:
        print('\n# MFMA pipe of the factorisation (SQ pass, last repetition): mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / '
              '(4 SQ_BUSY_CU_CYCLES) summed over the launches; wait / stall / active = fractions of SQ_WAVE_CYCLES')
        segs_q = blocks(load_pmc(opt['--sq']), lambda r: r['name'])
        for i in range(0, len(segs_q), REPS):
            N, dt = label(i)
            _, ps = segs_q[min(i + REPS - 1, len(segs_q) - 1)]
            acc = collections.defaultdict(lambda: collections.defaultdict(float))
            for r in ps:
                fam = short(r['name'])[:60]
                for c, v in r.items():
                    if c != 'name':
                        acc[fam][c] += v
                acc[fam]['n'] += 1
            tm = sum(k['SQ_VALU_MFMA_BUSY_CYCLES'] for k in acc.values())
            tb = sum(k['SQ_BUSY_CU_CYCLES'] for k in acc.values())
            print('%-7d %-4s all %d launches: mfma_util %.3f' % (N, dt, len(ps), tm / (4 * tb) if tb else 0))
            for fam, k in sorted(acc.items(), key=lambda kv: -kv[1]['SQ_BUSY_CU_CYCLES']):
                wc = k['SQ_WAVE_CYCLES'] or 1
                print('        %4d x %-60s mfma_util %.3f  wait_any %.2f  issue_stall %.2f  active %.2f  lds_conflict %.3f' % (
                    k['n'], fam, k['SQ_VALU_MFMA_BUSY_CYCLES'] / (4 * k['SQ_BUSY_CU_CYCLES']) if k['SQ_BUSY_CU_CYCLES'] else 0,
                    k['SQ_WAIT_ANY'] / wc, k['SQ_WAIT_INST_ANY'] / wc, k['SQ_ACTIVE_INST_ANY'] / wc,
                    k['SQ_LDS_BANK_CONFLICT'] / wc))


if __name__ == '__main__':
    main()
