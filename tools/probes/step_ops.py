#!/usr/bin/env python3
"""aten-level op census of ONE eager DSVI step of the bench workload (which torch glue ops remain around the HIP
kernels, with the Python source line that issued them): python tools/probes/step_ops.py"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'nonstationary-precip_amd'))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from torch.profiler import profile, ProfilerActivity  # noqa: E402


def main():
    dev = torch.device('cuda', 0)
    from nsgp.dist import PhiloxEps, dp_objective
    from nsgp.gp import settings
    x_all, y_all = bench.synthetic_grid()
    x_in, y_in = x_all[:bench.BATCH].to(dev), y_all[:bench.BATCH].to(dev)
    model, mll, opt = bench.build(dev, 1)
    eps = PhiloxEps(bench.SEED, row0=0, step_dev=opt.step_dev)
    model.train()

    def step():
        eps.start_step(0, row0=0)
        opt.zero_grad()
        loss = -dp_objective(mll, model(x_in), y_in, bench.BATCH, 1)
        loss.backward()
        opt.bucket.gather_grads()
        opt.step(gather=False)

    with settings.num_likelihood_samples(bench.S_SAMPLES), settings.eps_provider(eps):
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
            step()
            torch.cuda.synchronize()
    want = ('aten::add', 'aten::add_', 'aten::sum', 'aten::mul', 'aten::neg', 'aten::copy_', 'aten::fill_', 'aten::zero_',
            'aten::sub', 'aten::div', 'aten::cat', 'aten::softplus', 'aten::softplus_backward', 'aten::mm', 'aten::bmm',
            'aten::matmul', 'aten::exp', 'aten::to', 'aten::_to_copy', 'aten::clone', 'aten::mean', 'aten::sigmoid')
    rows = {}
    for ev in prof.events():
        if (ev.name in want or ev.name.startswith('aten::_foreach') or os.environ.get('ALL_OPS')) and ev.name.startswith('aten::') and ev.self_device_time_total > 0:
            src = [s for s in ev.stack if 'nonstationary-precip_amd' in s or 'bench.py' in s]
            where = src[0].split('nonstationary-precip_amd/')[-1] if src else ('autograd engine' if not ev.stack else ev.stack[0][-60:])
            key = (ev.name, str([tuple(x) for x in ev.input_shapes if x]), where)
            r = rows.setdefault(key, [0, 0.0])
            r[0] += 1
            r[1] += ev.self_device_time_total
    for (name, shapes, where), (n, t) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
        print(f'{name:24s} x{n:2d} {t:7.1f} us  {shapes:48s} {where}')


if __name__ == '__main__':
    main()
