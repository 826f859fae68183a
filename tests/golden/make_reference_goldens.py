#!/usr/bin/env python3
"""Generate golden vectors by importing the parts of the reference that import cleanly here.

Run in the build container only (needs /root/reference):

    python tests/golden/make_reference_goldens.py

What is run from the reference (torch/pandas only -- no gpytorch needed):
  * utils/functional.py : dot, t, mv (both branches), op           -> ref_functional.npz
  * utils/dataprep.py   : download_data, whitening_transform,
                          train_test_split on the bundled CSVs      -> ref_dataprep.npz
Everything under models/* needs gpytorch (absent, not installable) and is therefore NOT run;
those paths are "parity unpinned" by reference artefacts (see oracle/__init__.py).
The bundled CSVs themselves are copied verbatim to tests/golden/data/ (data, not source).
"""
import os
import sys
import numpy as np
import torch

REF = '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    sys.path.insert(0, REF)
    import utils.functional as rfn           # noqa: E402  (the reference's own module)
    import utils.dataprep as rdp             # noqa: E402

    g = torch.Generator().manual_seed(173)   # BASE_SEED, utils/config.py:16
    out = {}
    v1 = torch.randn(3, 5, 7, generator=g, dtype=torch.float64)
    v2 = torch.randn(3, 5, 7, generator=g, dtype=torch.float64)
    A = torch.randn(3, 7, 7, generator=g, dtype=torch.float64)
    A = A @ A.transpose(-1, -2) + 7 * torch.eye(7, dtype=torch.float64)
    b = torch.randn(3, 7, generator=g, dtype=torch.float64)
    e1 = torch.rand(2, 6, generator=g, dtype=torch.float64) + 0.1
    e2 = torch.rand(2, 4, generator=g, dtype=torch.float64) + 0.1
    out.update(v1=v1, v2=v2, A=A, b=b, e1=e1, e2=e2,
               dot=rfn.dot(v1, v2), t=rfn.t(A), mv=rfn.mv(A, b), mv_inv=rfn.mv(A, b, invert=True),
               op=rfn.op(e1, e2), op_self=rfn.op(e1))
    np.savez_compressed(os.path.join(HERE, 'ref_functional.npz'),
                        **{k: v.numpy() for k, v in out.items()})

    dp = {}
    for name in ('uib_spatial', 'khyber_time_series'):
        data = rdp.download_data(os.path.join(REF, 'data', name + '.csv'))
        x, y, mx, sx, my, sy = rdp.whitening_transform(data)
        trx, try_, tex, tey = rdp.train_test_split(x, y, 0.8)
        dp.update({f'{name}_x': x, f'{name}_y': y, f'{name}_meanx': mx, f'{name}_stdx': sx,
                   f'{name}_meany': my, f'{name}_stdy': sy, f'{name}_ntrain': torch.tensor(len(trx)),
                   f'{name}_train_x_tail': trx[-3:], f'{name}_test_y_head': tey[:3]})
    np.savez_compressed(os.path.join(HERE, 'ref_dataprep.npz'),
                        **{k: v.numpy() for k, v in dp.items()})
    print('wrote ref_functional.npz, ref_dataprep.npz')


if __name__ == '__main__':
    main()
