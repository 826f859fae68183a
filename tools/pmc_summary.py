#!/usr/bin/env python3
"""Summarise rocprofv3 `--pmc` passes: per kernel name, mean counter value per dispatch.
    python tools/pmc_summary.py <dir-with-*_counter_collection.csv> [more dirs ...] [--match substr]
FETCH_SIZE / WRITE_SIZE are reported in KiB by rocprofv3; on gfx950 FETCH_SIZE counts 64 B per 128-B request
for wide streaming reads (MI355X_MICROARCH.md, HBM section) -- the `x2` column applies that correction."""
import collections
import csv
import glob
import os
import re
import sys


def main():
    args = sys.argv[1:]
    match = None
    if '--match' in args:
        i = args.index('--match')
        match = args[i + 1]
        del args[i:i + 2]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in args:
        for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
            for r in csv.DictReader(open(f)):
                name = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name'])
                name = re.sub(r'^void ', '', name)
                name = re.sub(r'\(.*$', '', name)[:70]
                if match and match not in name:
                    continue
                key = (name, r.get('Grid_Size', r.get('Grid_Size_X', '')))
                acc[key][r['Counter_Name']].append(float(r['Counter_Value']))
    for (name, grid), ctr in sorted(acc.items()):
        parts = []
        for c, v in sorted(ctr.items()):
            m = sum(v) / len(v)
            if c == 'FETCH_SIZE':
                parts.append(f'{c}={m / 1024:.2f} MiB (x2 = {2 * m / 1024:.2f} MiB)')
            elif c == 'WRITE_SIZE':
                parts.append(f'{c}={m / 1024:.2f} MiB')
            else:
                parts.append(f'{c}={m:.4g}')
        n = len(next(iter(ctr.values())))
        print(f'{name}  grid={grid}  dispatches={n}\n    ' + '  '.join(parts))


if __name__ == '__main__':
    main()
