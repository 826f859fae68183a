"""world_size-2 gloo tests (CPU) of the data-parallel DSVI machinery in nsgp.dist / nsgp.optim:
sharding, the flat gradient bucket + one sum-all-reduce, and the identity that makes the scheme exact:

    sum_r [ (B_r/B) * mean_s sum_{i in r} ELL_si / B_r  -  KL / (G N) ]  ==  single-process ELBO

with the reparameterisation noise keyed by the GLOBAL row (Philox), so the union of the ranks' draws is
the single-GPU draw.  The per-rank objective values come from the CPU oracle (tests may use it as the
checker); the product code under test is the host logic, which is device-agnostic."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _toy_problem(seed=173, B=64, S=3, M=10, D=3):
    g = torch.Generator().manual_seed(seed)
    F = torch.float64
    hidden = dict(Z=torch.randn(2, M, D, generator=g, dtype=F), lengthscale=torch.rand(2, 1, D, generator=g, dtype=F) + 0.6,
                  outputscale=torch.rand(2, generator=g, dtype=F) + 0.5, m=0.2 * torch.randn(2, M, generator=g, dtype=F),
                  Lq=torch.tril(0.1 * torch.randn(2, M, M, generator=g, dtype=F)) + torch.eye(M, dtype=F),
                  mean=('linear', torch.randn(D, 1, generator=g, dtype=F), torch.randn(1, generator=g, dtype=F)))
    last = dict(Z=torch.randn(M, 2, generator=g, dtype=F), lengthscale=torch.rand(1, 2, generator=g, dtype=F) + 0.6,
                outputscale=torch.rand((), generator=g, dtype=F) + 0.5, m=0.2 * torch.randn(M, generator=g, dtype=F),
                Lq=torch.tril(0.1 * torch.randn(M, M, generator=g, dtype=F)) + torch.eye(M, dtype=F),
                mean=('constant', torch.zeros(1, dtype=F)))
    x = torch.randn(B, D, generator=g, dtype=F)
    y = torch.randn(B, generator=g, dtype=F)
    return hidden, last, x, y, S


def _worker(rank, world, port, out_dir):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, 'nonstationary-precip_amd')):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from nsgp.dist import shard_bounds, DataParallel
    from nsgp.optim import FlatBucket
    from oracle import svgp, philox

    hidden, last, x, y, S = _toy_problem()
    B, N = x.shape[0], 1000
    lo, hi = shard_bounds(B, world, rank)
    params = [hidden['Z'], hidden['m'], hidden['Lq'], last['Z'], last['m'], last['Lq']]
    leaves = [torch.nn.Parameter(p.clone()) for p in params]
    bucket = FlatBucket(leaves)
    dp = DataParallel(bucket)
    assert dp.world == world and dp.rank == rank
    # all parameters live in one flat buffer, gradients too
    assert all(p.data_ptr() >= bucket.flat_p.data_ptr() for p in leaves)
    h = dict(hidden, Z=leaves[0], m=leaves[1], Lq=leaves[2])
    l_ = dict(last, Z=leaves[3], m=leaves[4], Lq=leaves[5])
    eps = torch.from_numpy(philox.normal(173, (5 << 32) | 0, lo, S, hi - lo, 2))          # rows lo..hi-1
    mean, var = svgp.dgp_forward(x[lo:hi], h, l_, 1, [eps], S)
    ell = svgp.gauss_ell(y[lo:hi], mean, var, 0.3).sum(-1) / (hi - lo)
    kl = svgp.kl_whitened(h) + svgp.kl_whitened(l_)
    obj = (((hi - lo) / B) * ell - kl / N / world).mean(0)                                # nsgp.dist.dp_objective
    bucket.zero_grad()
    (-obj).backward()
    local = obj.detach().clone()
    dp.allreduce_grads()
    dist.all_reduce(local)
    if rank == 0:
        np.savez(os.path.join(out_dir, 'dp.npz'), obj=local.numpy(), grad=bucket.unpadded(bucket.flat_g).numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_shard_bounds_cover_and_balance():
    from nsgp.dist import shard_bounds
    for n in (4096, 315, 7, 1):
        for world in (1, 2, 3, 4, 8):
            spans = [shard_bounds(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_flat_bucket_views_and_zero_grad():
    from nsgp.optim import FlatBucket
    ps = [torch.nn.Parameter(torch.randn(3, 4)), torch.nn.Parameter(torch.randn(5)),
          torch.nn.Parameter(torch.randn(2, 2), requires_grad=False)]
    before = [p.detach().clone() for p in ps]
    b = FlatBucket(ps)
    assert b.num_param_elements == 17 and b.numel == 128       # every parameter starts on a 256-byte boundary
    assert all(off % 64 == 0 for off, _ in b.offsets)
    assert torch.equal(ps[0], before[0]) and torch.equal(ps[1], before[1])
    (ps[0].sum() * 2 + (ps[1] ** 2).sum()).backward()
    assert torch.allclose(b.flat_g[:12], torch.full((12,), 2.0))
    assert torch.allclose(b.flat_g[64:69], 2 * before[1])
    assert torch.allclose(b.unpadded(b.flat_g), torch.cat([torch.full((12,), 2.0), 2 * before[1]]))
    assert float(b.flat_g[12:64].abs().sum()) == 0.0 and float(b.flat_g[69:].abs().sum()) == 0.0      # padding stays zero
    b.zero_grad()
    assert float(b.flat_g.abs().sum()) == 0.0 and ps[0].grad.data_ptr() == b.flat_g.data_ptr()
    with torch.no_grad():
        b.flat_p.add_(1.0)                              # an optimiser step on the flat buffer moves the params
    assert torch.allclose(ps[0], before[0] + 1)


def test_two_rank_gloo_allreduce_reproduces_single_process_gradient(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    z = np.load(os.path.join(str(tmp_path), 'dp.npz'))
    # single-process reference on the full minibatch with the same global-row-keyed noise
    from oracle import svgp, philox
    hidden, last, x, y, S = _toy_problem()
    leaves = [p.clone().requires_grad_() for p in (hidden['Z'], hidden['m'], hidden['Lq'], last['Z'], last['m'], last['Lq'])]
    h = dict(hidden, Z=leaves[0], m=leaves[1], Lq=leaves[2])
    l_ = dict(last, Z=leaves[3], m=leaves[4], Lq=leaves[5])
    eps = torch.from_numpy(philox.normal(173, (5 << 32) | 0, 0, S, x.shape[0], 2))
    elbo = svgp.dsvi_elbo(x, y, h, l_, 1, [eps], S, 0.3, 1000)
    grads = torch.autograd.grad(-elbo, leaves)
    flat = torch.cat([g.reshape(-1) for g in grads]).numpy()
    assert abs(float(z['obj']) - float(elbo)) < 1e-12
    assert np.allclose(z['grad'], flat, rtol=1e-10, atol=1e-12)


def test_philox_rows_are_partition_invariant_on_cpu():
    from oracle import philox
    full = philox.normal(7, 3, 0, 2, 100, 3)
    parts = np.concatenate([philox.normal(7, 3, 0, 2, 37, 3), philox.normal(7, 3, 37, 2, 63, 3)], axis=1)
    assert np.array_equal(full, parts)


def _drain_worker(rank, world, port, out_dir):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, 'nonstationary-precip_amd')):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from nsgp.dist import DataParallel
    from nsgp.optim import FlatBucket
    ps = [torch.nn.Parameter(torch.full((70,), float(rank + 1))), torch.nn.Parameter(torch.full((5,), 10.0 * (rank + 1)))]
    bucket = FlatBucket(ps, stage_of={id(ps[0]): 0, id(ps[1]): 1})
    dp = DataParallel(bucket)
    bucket.zero_grad()
    (ps[0].sum() * (rank + 1) + ps[1].sum() * 2).backward()
    dp.check_drained()                                  # nothing started yet
    dp.allreduce_stage(0)
    dp.allreduce_stage(1)
    raised = False
    try:
        dp.check_drained()                              # two exchanges in flight: the optimiser step must not run
    except RuntimeError:
        raised = True
    dp.wait_stages()
    dp.check_drained()
    ok = raised and dp.stages_issued == 2 and dp.stages_waited == 2
    g0, g1 = ps[0].grad, ps[1].grad                    # views of the bucket (grads_as_views)
    ok = ok and torch.allclose(g0, torch.full((70,), 3.0)) and torch.allclose(g1, torch.full((5,), 4.0))
    if rank == 0:
        np.savez(os.path.join(out_dir, 'drain.npz'), ok=np.array(ok))
    dist.barrier()
    dist.destroy_process_group()


def test_every_issued_stage_exchange_must_be_waited_on_before_the_optimiser_step(tmp_path):
    """DataParallel counts the asynchronous per-stage all-reduces it starts and joins;
    `check_drained()` (called by bench.py right before the Adam step) raises while one is pending."""
    world, port = 2, _free_port()
    mp.spawn(_drain_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    assert bool(np.load(os.path.join(str(tmp_path), 'drain.npz'))['ok'])
