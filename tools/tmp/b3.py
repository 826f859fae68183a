import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'nonstationary-precip_amd'))
import torch, bench
dev = torch.device('cuda', 0)
t0 = time.time()
print('b3 ms', bench.b3_sparse_multivariate_step_ms(dev), 'setup+time', time.time() - t0)
