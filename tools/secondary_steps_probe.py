#!/usr/bin/env python3
"""The two secondary training steps of bench.py as a stand-alone program for rocprofv3 passes:
the float64 Gibbs exact-GP MAP step at N = 4096 (experiments/spatial_exp.py:197-210) and the
BASELINE configs[2]-shaped sparse multivariate Gibbs step (M = 512, 5,676 rows).

    python tools/secondary_steps_probe.py [map] [b3]
"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'nonstationary-precip_amd'))
import torch  # noqa: E402
import bench  # noqa: E402

if __name__ == '__main__':
    which = sys.argv[1:] or ['map', 'b3']
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(0)
    if 'map' in which:
        print('gibbs_map_step f64 (N=4096):', bench.gibbs_map_step_ms(dev, 4096), flush=True)
    if 'b3' in which:
        print('b3_sparse_multivariate_step f32:', bench.b3_sparse_multivariate_step_ms(dev), flush=True)
