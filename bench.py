#!/usr/bin/env python3
"""bench.py -- DSVI ELBO steps/sec of the 2-layer deep GP (BASELINE.json configs[3]) on N MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE B4): DeepGP(num_layers=1) = hidden 3->2 + last 2->1, M=1024 inducing,
S=10 likelihood samples, minibatch B=4096 of a synthetic N=100,000 spatio-temporal grid
(100 months x 1,000 cells, time-major, z-scored), float32 with float64 Kzz Cholesky and the whitened projection
A = L^-1 Kzx as an exact int8 digit-plane product (settings.whiten_matmul_i8; better than the float64-accumulated product it
replaces, which stands in for the reference's float64 solve), Adam lr 0.01.
One step = forward + ELBO + backward + (gradient all-reduce) + Adam update on one 4096-row minibatch that
is already resident in HBM.  With N > 1 (data parallel, nsgp/dist.py) the default is SURVEY 8e's split: ONE
4096-row minibatch is shared by the N ranks (4096/N rows each; `"scaling": "strong"`, value = iterations/s);
`--scaling weak` gives every rank its own 4096 rows (global batch 4096 N, value = N x iterations/s).  Either way the
ranks' objectives sum to the single-process ELBO of the global minibatch and the flat gradient bucket is summed
with one RCCL all-reduce.

Prints ONE JSON line on rank 0 (see README / DESIGN.md for the field definitions).  `dtype` "f32" names the arithmetic of the
dominant kernels and of every stored tensor of the layers; `roofline` prices the float32 GEMM family against the data-sheet
157.3 TFLOP/s and also reports what a register-only MFMA loop sustains on the box at hand (`sustained_mfma_measured`);
`f64acc_projection` / `i8_projection` are the same objects for the float64-accumulating and the int8 products of the step.
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, 'nonstationary-precip_amd'))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

M_INDUCING, S_SAMPLES, BATCH, N_DATA, SEED = 1024, 10, 4096, 100_000, 173
MFMA_F32_PEAK_TFLOPS = 157.3          # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD
MFMA_F64_PEAK_TFLOPS = 78.6           # v_mfma_f64_16x16x4_f64: half the f32 matrix rate
MFMA_BF16_PEAK_TFLOPS = 2500.0        # v_mfma_f32_32x32x16_bf16, dense (MI355X_MICROARCH.md)
MFMA_I8_PEAK_TOPS = 5000.0            # v_mfma_i32_32x32x32_i8: 2x the bf16 rate per clock (same guide)
I8_PLANE_PRODUCTS = 14                # digit-plane products per multiply-add of the int8 projection (csrc/gemm_i8.hip; 15 with 5 Kzx planes)
HBM_PEAK_GBS = 8000.0


def synthetic_grid(n_months=100, n_cells=1000, seed=SEED):
    """(t, lon, lat) rows, time-major like data/uib_spatio_temporal.csv; y = sin(2 pi t) g(lon,lat) + noise."""
    g = torch.Generator().manual_seed(seed)
    lon = torch.arange(40, dtype=torch.float32) * 0.25 + 72.25
    lat = torch.arange(25, dtype=torch.float32) * 0.25 + 31.0
    cells = torch.cartesian_prod(lon, lat)                                  # 1000 cells
    t = (2000.0 + (torch.arange(n_months, dtype=torch.float32) + 0.5) / 12.0)
    x = torch.cat([t.repeat_interleave(n_cells).unsqueeze(-1), cells.repeat(n_months, 1)], dim=-1)
    field = torch.sin(0.8 * (cells[:, 0] - 76.0)) * torch.cos(0.9 * (cells[:, 1] - 34.0)) + 1.5
    y = torch.sin(2 * math.pi * t).repeat_interleave(n_cells) * field.repeat(n_months) \
        + 0.1 * torch.randn(n_months * n_cells, generator=g)
    sx, mx = torch.std_mean(x, dim=-2)
    sy, my = torch.std_mean(y)
    return (x - mx) / sx, (y - my) / sy                                    # utils/dataprep.py:35-43


class GemmTimer:
    """HIP-event pairs around every GEMM launch on torch's current stream (the stream the kernels use)."""

    def __init__(self):
        self.records = []

    def __call__(self, fn, flops, dtype):
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        out = fn()
        e1.record()
        self.records.append((e0, e1, flops, dtype))
        return out

    def summary(self, dtype):
        torch.cuda.synchronize()
        recs = [r for r in self.records if r[3] == dtype]
        ms = sum(a.elapsed_time(b) for a, b, _, _ in recs)
        return ms, sum(r[2] for r in recs), len(recs)


def build(device, dp_world, stage_of_fn=None):
    """`stage_of_fn(model, mll)` -> {id(p): exchange group} (staged backward, nsgp/stages.py): lays the flat gradient
    bucket out by the backward stage that completes each parameter's gradient."""
    import models.dgps as dgps
    from nsgp.gp.mlls import DeepApproximateMLL, VariationalELBO
    from nsgp.optim import FusedAdam
    torch.manual_seed(SEED)
    model = dgps.DeepGP(1, (N_DATA, 3), num_inducing=M_INDUCING).to(device)
    mll = DeepApproximateMLL(VariationalELBO(model.likelihood, model, N_DATA))
    stage_of = stage_of_fn(model, mll) if stage_of_fn is not None else None
    opt = FusedAdam(model.parameters(), lr=0.01, capturable=True, grads_as_views=False, stage_of=stage_of)
    return model, mll, opt


def gemm_source_sha():
    """sha256 of the GEMM kernel source: PMC traffic figures are only valid for the code they were measured on."""
    import hashlib
    path = os.path.join(ROOT, 'nonstationary-precip_amd', 'csrc', 'gemm.hip')
    try:
        return hashlib.sha256(open(path, 'rb').read()).hexdigest()[:16]
    except OSError:
        return None


def gemm_traffic(world, share):
    """`traffic`: average L2<->fabric bytes per f32 GEMM launch of a step, from the committed rocprofv3 PMC passes of
    this same single-GPU workload (tools/gemm_traffic.py -> profiles/r03/gemm_traffic.json).  PMC counters cannot be
    read from inside the process, so this is the recorded measurement, not a live one; it is reported only while the
    GEMM source still hashes to what the passes were taken on (null otherwise, and for N > 1)."""
    path = os.path.join(ROOT, 'profiles', 'r03', 'gemm_traffic.json')
    if world != 1 or share != 1 or not os.path.exists(path):
        return {'traffic': None}
    try:
        d = json.load(open(path))
        if d.get('gemm_source_sha') != gemm_source_sha():
            return {'traffic': None, 'traffic_note': 'profiles/r03/gemm_traffic.json is stale (gemm.hip changed since the PMC passes)'}
        return {'traffic': round(float(d['bytes_per_launch'])), 'traffic_unit': 'bytes per launch (FETCH_SIZE x2 + '
                'WRITE_SIZE, incl. Infinity-Cache hits)', 'traffic_source': 'profiles/r03/gemm_traffic.json',
                'algorithmic_bytes_per_launch': d.get('algorithmic_bytes_per_launch')}
    except (OSError, ValueError, KeyError):
        return {'traffic': None}


def host_cores():
    """CPU threads this process may actually use: cgroup quota if set, else the affinity mask."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        pass
    return max(1, n)


def cpu_baseline(x, y, idx, seconds_budget=14.0, mirror=True):
    """The oracle's op sequence for the same model/minibatch on the host cores: forward + backward + Adam.
    mirror=True: gpytorch's own sequence (float32, float64 Cholesky/solve, per-sample Kzz recomputation in the last layer
    [recalled]); mirror=False: the same arithmetic with Kzz / Cholesky computed once per layer (the redundancy-free
    restatement, a stronger baseline)."""
    from oracle import svgp
    g = torch.Generator().manual_seed(SEED)
    M, D = M_INDUCING, 3
    sp = torch.nn.functional.softplus

    def leafs(shapes):
        return [torch.zeros(s).requires_grad_() for s in shapes]
    hZ = torch.randn(2, M, D, generator=g).requires_grad_()
    lZ = torch.randn(M, 2, generator=g).requires_grad_()
    h_rl, h_ro, l_rl, l_ro, rn, lc = leafs([(2, 1, D), (2,), (1, 2), (), (1,), (1,)])
    hm = (1e-3 * torch.randn(2, M, generator=g)).requires_grad_()
    lm = (1e-3 * torch.randn(M, generator=g)).requires_grad_()
    hL = torch.eye(M).repeat(2, 1, 1).requires_grad_()
    lL = torch.eye(M).clone().requires_grad_()
    hw, hb = torch.randn(D, 1, generator=g).requires_grad_(), torch.randn(1, generator=g).requires_grad_()
    params = [hZ, lZ, h_rl, h_ro, l_rl, l_ro, rn, lc, hm, lm, hL, lL, hw, hb]
    state, times = {}, []
    t_all = time.perf_counter()
    step = 0
    while True:
        rows = idx[step % len(idx)][:BATCH]
        xb, yb = x[rows], y[rows]
        eps = [torch.randn(S_SAMPLES, BATCH, 2, generator=g)]
        t0 = time.perf_counter()
        hidden = dict(Z=hZ, lengthscale=sp(h_rl), outputscale=sp(h_ro), m=hm, Lq=hL, mean=('linear', hw, hb))
        last = dict(Z=lZ, lengthscale=sp(l_rl), outputscale=sp(l_ro), m=lm, Lq=lL, mean=('constant', lc))
        loss = -svgp.dsvi_elbo(xb, yb, hidden, last, 1, eps, S_SAMPLES, sp(rn) + 1e-4, N_DATA, mirror=mirror)
        grads = torch.autograd.grad(loss, params)
        with torch.no_grad():
            new = svgp.adam_step([p.detach() for p in params], list(grads), state)
            for p, q in zip(params, new):
                p.copy_(q)
        times.append(time.perf_counter() - t0)
        step += 1
        if step >= 2 and (time.perf_counter() - t_all > seconds_budget or step >= 5):
            break
    use = times[1:] if len(times) > 1 else times
    return len(use) / sum(use), len(times)


def _timeit(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def _eager_and_graphed_ms(step, reps=3, capture=True):
    """(eager ms, graph-replay ms or None, note): the step launched kernel by kernel, then captured once as a hipGraph and
    replayed (its optimiser must be capturable; host-side Cholesky checks are skipped while capturing)."""
    eager = _timeit(step, reps=reps)
    if not capture:
        return round(eager, 3), None, None
    try:
        from nsgp.graph import GraphedCallable
        g = GraphedCallable(step, warmup=1)
        return round(eager, 3), round(_timeit(g, reps=reps), 3), None
    except Exception as e:                               # a host synchronisation inside the step, an uncapturable torch op ...
        torch.cuda.synchronize()
        return round(eager, 3), None, repr(e)[:200]


def _b2_inputs(n, device, dt):
    """BASELINE B2 inputs: N = 394 is the real data/uib_spatial.csv (z-scored lon/lat); otherwise a regular lattice,
    z-scored; ell = exp(0.3 N(0,1) + log 0.3) in the reference's (D, N) layout."""
    g = torch.Generator().manual_seed(SEED)
    if n == 394:
        import pandas as pd
        d = pd.read_csv(os.path.join(ROOT, 'tests', 'golden', 'data', 'uib_spatial.csv'))
        x = torch.tensor(d[['lon', 'lat']].values, dtype=torch.float64)
    else:
        side = int(round(n ** 0.5))
        gx, gy = torch.meshgrid(torch.arange(side, dtype=torch.float64), torch.arange(n // side, dtype=torch.float64),
                                indexing='ij')
        x = torch.stack([gx.reshape(-1), gy.reshape(-1)], -1)
    x = (x - x.mean(0)) / x.std(0)
    ell = torch.exp(0.3 * torch.randn(2, x.shape[0], generator=g, dtype=torch.float64) + math.log(0.3))
    return x.to(device=device, dtype=dt), ell.to(device=device, dtype=dt)


def build_chol_table(device, with_cpu=True):
    """Second half of the BASELINE metric (B2): Gibbs K build + potrf(sigma_f^2 K + sigma^2 I) at
    N in {394 (real CSV), 1024, 4096, 16384}, D = 2, float32 and float64, with GB/s (algorithmic bytes of the build) and
    TFLOP/s (N^3/3) -- and, beside each, the CPU oracle timed on the host cores: the reference's own op sequence for the
    build (oracle.kernels.gibbs, 8 full-size temporaries) and torch.linalg.cholesky.  At N = 16384 the CPU build is timed
    on a 1024-row slab and scaled (its temporaries would need 6 x 4.3 GB); the CPU Cholesky is the full matrix."""
    from nsgp import ops
    from oracle import kernels as OK
    rows = []
    # GPU phase first, back to back (host-side oracle timings in between would let the GPU clocks fall back to idle:
    # the first version of this table read 8 ms for the 2.4 ms float64 potrf at N = 4096)
    spin = torch.empty(64 << 20, device=device)
    for _ in range(200):
        spin.add_(1.0)                                            # ~60 ms of streaming work: clocks up
    for n in (394, 1024, 4096, 16384):
        for dt, tag in ((torch.float32, 'f32'), (torch.float64, 'f64')):
            x, e = _b2_inputs(n, device, dt)
            nn = x.shape[0]
            os_ = torch.tensor([0.644], dtype=dt, device=device)
            nz = torch.tensor([0.011], dtype=dt, device=device)
            t_build = _timeit(lambda: ops.gibbs_build(x, x, e, e, os_, nz), reps=50 if nn <= 4096 else 10)
            K = ops.gibbs_build(x, x, e, e, os_, nz)
            t_chol = _timeit(lambda: ops.potrf(K), reps=20 if nn <= 4096 else 3)
            _, info = ops.potrf(K)
            bytes_ = K.element_size() * (nn * nn + 2 * 2 * (nn + nn))
            rows.append({'N': nn, 'dtype': tag, 'build_ms': round(t_build, 4), 'build_GBs': round(bytes_ / t_build / 1e6, 1),
                         'build_frac_hbm': round(bytes_ / t_build / 1e6 / HBM_PEAK_GBS, 3),
                         'potrf_ms': round(t_chol, 4), 'potrf_TFLOPs': round(nn ** 3 / 3 / t_chol / 1e9, 3),
                         'potrf_info': int(info.max().item())})
            del K
    del spin
    if with_cpu:
        for row in rows:
            nn, dt = row['N'], (torch.float32 if row['dtype'] == 'f32' else torch.float64)
            x, e = _b2_inputs(nn if nn != 394 else 394, 'cpu', dt)
            xc, ec = x, e
            slab = min(nn, 1024 if nn > 4096 else nn)
            reps = 3 if nn <= 1024 else 1
            t0 = time.perf_counter()
            for _ in range(reps):
                Kc = 0.644 * OK.gibbs(xc[:slab], xc, ec[:, :slab], ec)
            t_cb = (time.perf_counter() - t0) / reps * (nn / slab)
            del Kc
            Kfull = (0.644 * OK.gibbs(xc, xc, ec, ec) if nn <= 4096 else None)
            if Kfull is None:          # N = 16384: a well-conditioned SPD stand-in of the same size for the CPU Cholesky
                gq = torch.Generator().manual_seed(1)
                Q = torch.randn(nn, 64, generator=gq, dtype=dt)
                Kfull = Q @ Q.T
            Kfull.diagonal().add_(0.011 if nn <= 4096 else 1.0)
            t0 = time.perf_counter()
            torch.linalg.cholesky(Kfull)
            t_cc = time.perf_counter() - t0
            del Kfull
            row.update(cpu_build_ms=round(t_cb * 1e3, 2), cpu_potrf_ms=round(t_cc * 1e3, 2),
                       cpu_build_sample=('full' if slab == nn else f'{slab}-row slab, scaled'))
    return rows


def b3_sparse_multivariate_step_ms(device):
    """BASELINE B3: one training step (objective, backward, Adam) of the inducing-point GP over
    SparseMultivariateGibbsKernel, M = 512, on the 5,676 rows of uib_spatio_temporal.csv, float32
    (tests/test_gpu_cfg2.py holds the parity of this model)."""
    import pandas as pd
    import nsgp.gp as gpytorch
    from sklearn.cluster import KMeans
    from models.sparse_multivariate_gibbs_kernel import SparseMultivariateGibbsKernel
    d = pd.read_csv(os.path.join(ROOT, 'tests', 'golden', 'data', 'uib_spatio_temporal.csv'))
    xy = torch.tensor(d[['lon', 'lat']].values, dtype=torch.float32)
    y = torch.tensor(d['tp'].values, dtype=torch.float32)
    sx, mx = torch.std_mean(xy, dim=0)
    sy, my = torch.std_mean(y)
    x, y = ((xy - mx) / sx).to(device), ((y - my) / sy).to(device)
    Z = torch.tensor(KMeans(512, n_init=1, random_state=SEED).fit(x.cpu().numpy()).cluster_centers_, dtype=torch.float32)
    Z = (Z + 0.05 * torch.randn(Z.shape, generator=torch.Generator().manual_seed(0))).to(device)

    class SparsePSGP(gpytorch.models.ExactGP):
        def __init__(self, train_x, train_y, likelihood):
            super().__init__(train_x, train_y, likelihood)
            self.mean_module = gpytorch.means.ZeroMean()
            base = gpytorch.kernels.ScaleKernel(SparseMultivariateGibbsKernel(Z, 2, Z.clone()))
            self.covar_module = gpytorch.kernels.InducingPointKernel(base, inducing_points=Z.clone(), likelihood=likelihood)
            self.covar_module.inducing_points.requires_grad = False

        def forward(self, xx):
            return gpytorch.distributions.MultivariateNormal(self.mean_module(xx), self.covar_module(xx))
    torch.manual_seed(3)
    lik = gpytorch.likelihoods.GaussianLikelihood()
    model = SparsePSGP(x, y, lik).to(device)
    model.train(); lik.train()
    mll = gpytorch.mlls.ExactMarginalLogLikelihood(lik, model)
    from nsgp.optim import FusedAdam
    opt = FusedAdam([p for p in model.parameters() if p.requires_grad], lr=0.01, capturable=True)

    def step():
        opt.zero_grad()
        loss = -mll(model(model.train_inputs[0]), model.train_targets)
        loss.backward()
        opt.step()
    # not captured: the matrix-normal prior's conditional mean (models/sparse_multivariate_gibbs_kernel.py:65-75) goes through
    # host-side pieces (a 2 x 2 inverse on the CPU, jitter retries that read `info`); capturing it crashed the process
    eager, graphed, note = _eager_and_graphed_ms(step, capture=False)
    return {'eager_ms': eager, 'hipgraph_ms': graphed, 'optimizer': 'FusedAdam',
            'hipgraph_note': 'not captured (host-side pieces in the prior conditional mean)'}


def mfma_sustained(device):
    """What the matrix cores of THIS chip deliver with nothing but MFMAs in flight (nsgp_mfma_rate_probe: a chip-filling grid
    of register-only MFMA loops): {'f32' | 'f64' | 'i8': {'rate': TFLOP/s or TOP/s, 'clock_GHz': shader clock held under that
    load}}.  The data-sheet peaks the `roofline` objects are priced against assume 2.4 GHz; under full matrix-core load the
    chip holds ~2.1 GHz, so a GEMM that kept every matrix core busy all the time would reach rate / peak ~ 0.88, not 1."""
    import numpy as np
    from nsgp import _lib, ops
    out = {}
    wgs = 2048
    buf = torch.zeros(8 * wgs, dtype=torch.int64, device=device)
    sink = torch.zeros(4, dtype=torch.float32, device=device)
    for name, kind, iters, ops_per in (('f32', 0, 400, 4096), ('f64', 1, 400, 2048), ('i8', 2, 800, 65536)):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = None
        for rep in range(3):
            e0.record()
            _lib.call('nsgp_mfma_rate_probe', kind, wgs, iters, ops._p(buf), ops._p(sink), ops._stream())
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1)
            best = ms if best is None or ms < best else best
        a = buf.cpu().numpy().reshape(-1, 2).astype(np.float64)
        ok = a[:, 1] > 0
        clk = float(np.median(a[ok, 0] / a[ok, 1]) * 0.1) if ok.any() else None          # cycles per 10 ns tick -> GHz
        out[name] = {'rate': round(wgs * 4 * iters * 8 * ops_per / (best * 1e-3) / 1e12, 1),
                     'clock_GHz': round(clk, 3) if clk else None}
    return out


def gibbs_chol_ms(device, with_cpu=True):
    """BASELINE B2 table + the N = 4096 figures as flat fields (the metric's second half) + the float64 MAP step."""
    table = build_chol_table(device, with_cpu=with_cpu)
    out = {'build_chol': table}
    for row in table:
        if row['N'] == 4096:
            tag = row['dtype']
            out[f'gibbs_build_ms_{tag}'] = row['build_ms']
            out[f'gibbs_build_GBs_{tag}'] = row['build_GBs']
            out[f'potrf_ms_{tag}'] = row['potrf_ms']
            out[f'potrf_TFLOPs_{tag}'] = row['potrf_TFLOPs']
    m = gibbs_map_step_ms(device, 4096)
    out['gibbs_map_step_ms_f64'] = m['hipgraph_ms'] if m['hipgraph_ms'] is not None else m['eager_ms']
    out['gibbs_map_step_f64'] = m
    b3 = b3_sparse_multivariate_step_ms(device)
    out['b3_sparse_multivariate_step_ms_f32'] = b3['hipgraph_ms'] if b3['hipgraph_ms'] is not None else b3['eager_ms']
    out['b3_sparse_multivariate_step_f32'] = b3
    return out


def gibbs_map_step_ms(device, n):
    """One MAP training step of the Gibbs-kernel exact GP at N = n (BASELINE configs[1] on a synthetic lattice,
    SURVEY 8d cfg2; the flow of experiments/spatial_exp.py:197-210 in float64): kernel build, log-normal prior
    log-density of the lengthscale field (2 more N x N Cholesky factorisations), marginal likelihood, backward, Adam."""
    import nsgp.gp as gpytorch
    from models.gibbs_kernels import LogNormalPriorProcess
    from models.nonstationary_models import DiagonalExactGP
    side = int(round(n ** 0.5))
    gx, gy = torch.meshgrid(torch.arange(side, dtype=torch.float64), torch.arange(n // side, dtype=torch.float64),
                            indexing='ij')
    x = torch.stack([gx.reshape(-1), gy.reshape(-1)], -1)
    x = (x - x.mean(0)) / x.std(0)
    y = torch.sin(3.0 * x[:, 0]) * torch.cos(2.0 * x[:, 1]) + 0.1 * torch.randn(x.shape[0], dtype=torch.float64,
                                                                                    generator=torch.Generator().manual_seed(SEED))
    prior = LogNormalPriorProcess(input_dim=2).double().to(device)
    prior.covar_module.base_kernel.lengthscale = 1.3 * torch.ones_like(prior.covar_module.base_kernel.lengthscale)
    prior.mean_module.constant = torch.nn.Parameter(math.log(0.3) * torch.ones_like(prior.mean_module.constant))
    for p in prior.parameters():
        p.requires_grad = False
    lik = gpytorch.likelihoods.GaussianLikelihood().double()
    model = DiagonalExactGP(x, y, lik, prior, num_dim=2).double().to(device)
    model.likelihood.noise = 0.011
    model.covar_module.outputscale = 0.644
    for p in list(model.likelihood.noise_covar.parameters()) + [model.covar_module._parameters['raw_outputscale']]:
        p.requires_grad = False
    model.train()
    lik.train()
    mll = gpytorch.mlls.ExactMarginalLogLikelihood(lik, model)
    xd, yd = model.train_inputs[0], model.train_targets
    # float64 model: FusedAdam keeps float32 buckets, so this step uses torch's own capturable Adam
    opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=0.01, capturable=True, foreach=True)

    def step():
        opt.zero_grad(set_to_none=True)
        loss = -mll(model(xd), yd)
        loss.backward()
        opt.step()
    eager, graphed, note = _eager_and_graphed_ms(step)
    return {'eager_ms': eager, 'hipgraph_ms': graphed, **({'hipgraph_note': note} if note else {})}


def cpu_baseline_cfg5(rows=1024, M=2048, S=10, n_data=1_000_000, seconds_budget=30.0):
    """The oracle's op sequence for BASELINE configs[4]'s model (3-layer = tied 3->3 hidden layer applied twice + last
    3->1, M = 2048, S = 10) on the host cores: forward + backward + Adam in float32 with float64 Cholesky, Kzz / Cholesky
    once per layer (the redundancy-free restatement; gpytorch-mirror mode would recompute them per sample).  A BOUNDED
    sample: minibatch of `rows` rows instead of 4096 (a full step is minutes of CPU work).  Returns (seconds per sampled
    step, steps timed)."""
    from oracle import svgp
    g = torch.Generator().manual_seed(SEED)
    D = 3
    sp = torch.nn.functional.softplus
    hZ = torch.randn(3, M, D, generator=g).requires_grad_()
    lZ = torch.randn(M, 3, generator=g).requires_grad_()
    h_rl, h_ro, l_rl, l_ro, rn, lc = [torch.zeros(sh).requires_grad_() for sh in [(3, 1, D), (3,), (1, 3), (), (1,), (1,)]]
    hm = (1e-3 * torch.randn(3, M, generator=g)).requires_grad_()
    lm = (1e-3 * torch.randn(M, generator=g)).requires_grad_()
    hL = torch.eye(M).repeat(3, 1, 1).requires_grad_()
    lL = torch.eye(M).clone().requires_grad_()
    hw, hb = torch.randn(D, 1, generator=g).requires_grad_(), torch.randn(1, generator=g).requires_grad_()
    params = [hZ, lZ, h_rl, h_ro, l_rl, l_ro, rn, lc, hm, lm, hL, lL, hw, hb]
    state, times = {}, []
    t_all = time.perf_counter()
    while True:
        xb, yb = torch.randn(rows, D, generator=g), torch.randn(rows, generator=g)
        eps = [torch.randn(S, rows, 3, generator=g) for _ in range(2)]
        t0 = time.perf_counter()
        hidden = dict(Z=hZ, lengthscale=sp(h_rl), outputscale=sp(h_ro), m=hm, Lq=hL, mean=('linear', hw, hb))
        last = dict(Z=lZ, lengthscale=sp(l_rl), outputscale=sp(l_ro), m=lm, Lq=lL, mean=('constant', lc))
        loss = -svgp.dsvi_elbo(xb, yb, hidden, last, 2, eps, S, sp(rn) + 1e-4, n_data, mirror=False)
        grads = torch.autograd.grad(loss, params)
        with torch.no_grad():
            for p_, q in zip(params, svgp.adam_step([p_.detach() for p_ in params], list(grads), state)):
                p_.copy_(q)
        times.append(time.perf_counter() - t0)
        if len(times) >= 2 and (time.perf_counter() - t_all > seconds_budget or len(times) >= 3):
            break
        if len(times) == 1 and time.perf_counter() - t_all > seconds_budget:
            break
    use = times[1:] if len(times) > 1 else times
    return sum(use) / len(use), len(times)


def main_cfg5(args):
    """BASELINE configs[4] on one GPU as a bench.py line (single process; the 8-GPU form of this config shares the
    data-parallel path of the headline config): `roofline` for the f32 GEMM family (+ the float64-accumulating and bf16
    projection families), `cpu_baseline` from the oracle on a bounded sample."""
    if int(os.environ.get('WORLD_SIZE', '1')) != 1:
        raise SystemExit('--config cfg5 is a single-GPU run')
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import dsvi_cfg5_probe as probe
    torch.cuda.set_device(0)
    pa = probe.parser().parse_args(['--steps', str(args.steps), '--warmup', str(args.warmup), '--forward', args.forward])
    r = probe.run(pa)
    ach = r['f32_gemm_TFLOPs']
    line = {'metric': 'dsvi_elbo_steps_per_sec', 'value': r['steps_per_sec'], 'unit': 'steps/s', 'n_gpus': 1,
            'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': r['ms_per_step'], 'higher_is_better': True,
            'scaling': 'strong', 'vs_baseline': None,
            'dtype': 'f32' if args.forward == 'f32' else 'f32 (bf16 forward projections: ' + args.forward + ')',
            'data': 'synthetic', 'config': {'workload': r['workload'], 'M': 2048, 'S': 10, 'global_batch': 4096,
                                            'N': 1000000, 'parallelism': 'dp1', 'forward': args.forward},
            'roofline': {'bound': 'mfma', 'kernel': 'gemm_kernel<float,...> (all f32 GEMM launches of a step)',
                         'achieved': ach, 'peak': MFMA_F32_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                         'frac': round(ach / MFMA_F32_PEAK_TFLOPS, 4), 'traffic': None,
                         'gemm_ms_per_step': r['f32_gemm_ms_per_step'], 'gemm_launches_per_step': r['f32_gemm_launches']},
            'f64acc_projection': ({'ms_per_step': r['f64acc_gemm_ms_per_step'], 'launches_per_step': r['f64acc_gemm_launches'],
                                   'achieved': r['f64acc_gemm_TFLOPs'], 'peak': MFMA_F64_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                                   'frac': round(r['f64acc_gemm_TFLOPs'] / MFMA_F64_PEAK_TFLOPS, 4)}
                                  if r['f64acc_gemm_launches'] else None),
            'i8_projection': ({'ms_per_step': r['i8_gemm_ms_per_step'], 'launches_per_step': r['i8_gemm_launches'],
                               'achieved': round(I8_PLANE_PRODUCTS * r['i8_gemm_TFLOPs_f64eq'] / 1e0, 1), 'peak': MFMA_I8_PEAK_TOPS,
                               'unit': 'TOP/s (int8)', 'frac': round(I8_PLANE_PRODUCTS * r['i8_gemm_TFLOPs_f64eq'] / MFMA_I8_PEAK_TOPS, 4),
                               'f64_equivalent_TFLOPs': r['i8_gemm_TFLOPs_f64eq']} if r.get('i8_gemm_launches') else None),
            'bf16_projection': ({'ms_per_step': r['bf16_gemm_ms_per_step'], 'launches_per_step': r['bf16_gemm_launches'],
                                 'achieved': r['bf16_gemm_TFLOPs'], 'peak': MFMA_BF16_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                                 'frac': round(r['bf16_gemm_TFLOPs'] / MFMA_BF16_PEAK_TFLOPS, 4)}
                                if r['bf16_gemm_launches'] else None),
            'detail': r}
    if not args.no_cpu_baseline:
        torch.set_num_threads(host_cores())
        rows = 1024
        sec, nsteps = cpu_baseline_cfg5(rows=rows)
        # per-step CPU time is dominated by the O(M^2 n) projections (n proportional to the minibatch) plus the O(M^3)
        # Cholesky chain; reported as steps/s of the SAMPLED minibatch and, scaled by rows / 4096, as a full-minibatch figure
        line['cpu_baseline'] = {'value': round(1.0 / sec * rows / BATCH, 5), 'unit': 'steps/s', 'cores': torch.get_num_threads(),
                                'kind': 'port',
                                'sample': f'{nsteps} DSVI steps (first discarded when more than one) of the configs[4] model '
                                          f'(M=2048, S=10, 3 layers) on a {rows}-row minibatch, oracle with Kzz + Cholesky once '
                                          f'per layer; value = measured {1.0 / sec:.4f} steps/s x {rows}/{BATCH} (linear in the '
                                          'minibatch: an upper bound on the CPU rate, the M^3 part does not shrink)',
                                'sampled_steps_per_sec': round(1.0 / sec, 5), 'sampled_rows': rows}
        line['speedup_vs_cpu'] = round(line['value'] / line['cpu_baseline']['value'], 1)
    print(json.dumps(line), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=100)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--config', choices=('cfg4', 'cfg5'), default='cfg4',
                    help='cfg4 (default): BASELINE configs[3], the headline 2-layer M=1024 workload; cfg5: configs[4], the 3-layer '
                         'M=2048 N=1e6 shape on ONE GPU (tools/dsvi_cfg5_probe.py), see --forward')
    ap.add_argument('--forward', choices=('f32', 'bf16', 'bf16_all'), default='f32',
                    help="cfg5 only: settings.forward_precision -- configs[4]'s bf16 forward")
    ap.add_argument('--no-graph', action='store_true', help='launch every kernel eagerly (no hipGraph replay)')
    ap.add_argument('--no-build-chol', action='store_true',
                    help='skip the Gibbs-build + Cholesky fields (PMC passes of the DSVI step only)')
    ap.add_argument('--scaling', choices=('weak', 'strong'), default='strong',
                    help='strong (default, SURVEY 8e): ONE 4096-row minibatch is split over the N GPUs (4096/N rows each), '
                         'value = iterations/s; weak: every GPU takes its own 4096-row minibatch per iteration (global batch '
                         '4096 x N), value = N x iterations/s.  Identical at N=1.')
    ap.add_argument('--split-graph', action='store_true',
                    help='analysis only (N=1): replay the step as the TWO graphs an N>1 run uses (forward+backward, then Adam) '
                         'with the eager gradient all-reduce call between them (a no-op at N=1), to price the split')
    ap.add_argument('--no-overlap', action='store_true',
                    help='N>1: ONE all-reduce of the whole gradient bucket after the backward pass (round-1 behaviour) instead '
                         'of the staged backward whose per-stage all-reduces overlap with the rest of the backward')
    ap.add_argument('--staged', action='store_true',
                    help='analysis only (N=1): run the step through the staged backward of an N>1 run (one graph per exchange '
                         'group, the all-reduces are no-ops at N=1), to price the staging')
    ap.add_argument('--rehearse-rccl', action='store_true',
                    help='analysis only (N=1): create a ONE-rank RCCL process group and issue every collective of the N>1 step '
                         'through it (staged backward, asynchronous per-stage all-reduces between the graph replays), so the '
                         'ProcessGroupNCCL / stream / hipGraph interplay is exercised on a one-GPU box')
    ap.add_argument('--fp32-whiten', action='store_true',
                    help="analysis only: round 1's arithmetic -- the whitened projection A = L^-1 Kzx as an fp32 product "
                         '(settings.whiten_matmul_f64(False)); misses the 1e-4 posterior-mean bound at this shape')
    ap.add_argument('--fuse-kzx', action='store_true',
                    help='analysis only: generate the Kzx tiles inside the loader of the forward projection instead of '
                         'materialising Kzx (settings.fuse_kzx(True)): less HBM traffic, slower product')
    ap.add_argument('--rank-share', type=int, default=1, metavar='G',
                    help='analysis only: run ONE rank\'s share of a G-rank job on this GPU (rows [0, 4096/G) of every '
                         'minibatch, the objective scaled as on rank 0 of G; no collective).  The JSON line is marked '
                         '"analysis" and is not a result for --gpus G.')
    args = ap.parse_args()

    if args.config == 'cfg5':
        return main_cfg5(args)
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU: the nsgp HIP backend has no CPU fallback')
    ndev = torch.cuda.device_count()
    dev_index = local_rank % max(ndev, 1)          # (rehearsals on a 1-GPU box put every rank on cuda:0)
    torch.cuda.set_device(dev_index)
    device = torch.device('cuda', dev_index)
    rehearse = args.rehearse_rccl and world == 1
    if rehearse:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29555')
        os.environ['RANK'], os.environ['WORLD_SIZE'] = '0', '1'
    if world > 1 or rehearse:
        backend = os.environ.get('NSGP_DIST_BACKEND', 'nccl')              # 'nccl' is RCCL on ROCm
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=device)
        else:
            dist.init_process_group(backend)

    from nsgp import ops
    from nsgp.dist import DataParallel, PhiloxEps, dp_objective, shard_bounds
    from nsgp.gp import settings
    from nsgp.gp.module import transform_cache

    x_all, y_all = synthetic_grid()
    gperm = torch.Generator().manual_seed(SEED)
    perm = torch.randperm(N_DATA, generator=gperm)
    weak = args.scaling == 'weak' and world > 1
    gbatch = BATCH * world if weak else BATCH                     # rows of one global minibatch
    n_batches = N_DATA // gbatch
    idx = [perm[i * gbatch:(i + 1) * gbatch] for i in range(n_batches)]     # shared shuffled index
    share = max(1, args.rank_share)
    if share > 1 and (weak and world > 1):
        raise SystemExit('--rank-share is a single-process analysis of the strong-scaling split')
    lo, hi = shard_bounds(gbatch, world * share, rank)            # share > 1: analysis mode (see --rank-share)
    xs = [x_all[i[lo:hi]].to(device) for i in idx]                           # resident in HBM
    ys = [y_all[i[lo:hi]].to(device) for i in idx]

    # Staged backward (N > 1): the gradient exchange of a stage's parameters runs under the next stage's kernels
    staged = (world > 1 and not args.no_overlap) or args.staged or rehearse
    plan, groups = None, {}
    MAX_GROUPS = 3                # last layer | hidden layers | whitening chain + hyper-parameters (a few KB)

    def discover_stages(model, mll):
        # one forward pass with the cuts in place; the assignment depends on the model's structure only
        model.train()
        with settings.num_likelihood_samples(S_SAMPLES), settings.eps_provider(PhiloxEps(SEED, row0=lo)), \
                settings.backward_stages(plan), transform_cache():
            loss = dp_objective(mll, model(xs[0]), ys[0], gbatch, world * share, negate=True)
        params = [p for p in model.parameters() if p.requires_grad]
        final = plan.final_stage_of(loss, params)
        groups['n'] = min(plan.num_stages, MAX_GROUPS)
        groups['stages'] = plan.num_stages
        return {pid: min(k, groups['n'] - 1) for pid, k in final.items()}

    if staged:
        from nsgp.stages import BackwardStages
        plan = BackwardStages()
    model, mll, opt = build(device, world, discover_stages if staged else None)
    dp = DataParallel(opt.bucket, force=rehearse)
    dp.broadcast_params()
    eps = PhiloxEps(SEED, row0=lo, step_dev=opt.step_dev)     # step counter lives on the device
    model.train()
    # static minibatch buffers: a step always reads these (hipGraph replays need fixed addresses)
    x_in, y_in = torch.empty_like(xs[0]), torch.empty_like(ys[0])

    one = torch.ones((), device=device)

    def fwd_bwd():
        eps.start_step(0, row0=lo)
        opt.zero_grad()
        with transform_cache():                  # the likelihood's noise shares the model's packed softplus launch
            out = model(x_in)
            loss = dp_objective(mll, out, y_in, gbatch, world * share, negate=True)    # = -(rank's share of the ELBO)
        loss.backward(gradient=one)          # resident seed: no ones_like fill launch per step
        opt.bucket.gather_grads()            # one multi-tensor copy into the flat gradient bucket
        return loss.detach()

    def adam_step():
        dp.check_drained()                   # every gradient exchange that was started has been waited on
        opt.step(gather=False)

    held = {}
    staged_graph_error = None

    def stage_group_fn(gi):
        # exchange group gi of the staged backward: group 0 = forward + ELBO + the loss's own backward stage
        def fn():
            if gi == 0:
                eps.start_step(0, row0=lo)
                opt.zero_grad()
                with settings.backward_stages(plan), transform_cache():
                    out = model(x_in)
                    held['loss'] = dp_objective(mll, out, y_in, gbatch, world * share, negate=True)
                plan.run_stage(0, held['loss'], one)
            for k in range(1, plan.num_stages):
                if min(k, groups['n'] - 1) == gi:
                    plan.run_stage(k)
            opt.bucket.gather_grads(gi)              # this group's gradients -> their contiguous range of the bucket
            return held['loss'].detach() if gi == 0 else None
        return fn

    def exchange_fn(gi):
        def fn():
            dp.allreduce_stage(gi, gather=False)     # asynchronous: the next group's kernels run under it
            if gi == groups['n'] - 1:
                dp.wait_stages()
        return fn

    def whole_step():
        loss = fwd_bwd()
        adam_step()
        return loss

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    with settings.num_likelihood_samples(S_SAMPLES), settings.eps_provider(eps), \
            settings.whiten_matmul_f64(not args.fp32_whiten), settings.fuse_kzx(args.fuse_kzx):
        use_graph = not args.no_graph
        x_in.copy_(xs[0]); y_in.copy_(ys[0])
        with torch.no_grad():
            model(x_in)                                  # first call draws the N(0, 1e-3^2) variational-mean init
        dp.broadcast_params()
        if use_graph:
            from nsgp.graph import GraphedCallable
            p0 = opt.bucket.flat_p.detach().clone()      # graph warm-up / capture runs real steps: undo them below
            if staged:
                from nsgp.graph import GraphedSequence
                capture_ok = 1
                try:
                    g_seq = GraphedSequence([stage_group_fn(gi) for gi in range(groups['n'])],
                                            [exchange_fn(gi) for gi in range(groups['n'])])
                    # test hook (tests/test_gpu_dist.py): a failure on ONE rank, after the sequence's collectives have been
                    # issued identically everywhere (a rank that dies before them desynchronises the group whatever we do)
                    if os.environ.get('NSGP_BENCH_FAIL_CAPTURE', '') in ('all', str(rank)):
                        raise RuntimeError(f'NSGP_BENCH_FAIL_CAPTURE: simulated capture failure on rank {rank}')
                    g_adam = GraphedCallable(adam_step, warmup=1)
                except Exception as e:           # e.g. a collective backend that does not tolerate the capture sequence
                    capture_ok = 0
                    staged_graph_error = repr(e)[:300]
                # The choice between graph replay and eager launches is made COLLECTIVELY (MIN over the ranks): a failure on
                # one rank only must not leave the ranks on different paths.  Either path issues the same collectives in
                # the same order; the eager path is as fast at this size (5.11 vs 5.09 ms at N = 1) and still overlapped.
                if world > 1 or rehearse:
                    flag = torch.tensor([capture_ok], dtype=torch.int32, device=device)
                    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                    capture_all = int(flag.item())
                else:
                    capture_all = capture_ok
                if not capture_all:
                    use_graph = False
                    if staged_graph_error is None:
                        staged_graph_error = 'capture failed on another rank'
                    try:                         # a collective error surfaces here: never run on after it
                        dp.wait_stages()
                        torch.cuda.synchronize()
                    except Exception as e2:
                        print(f'bench.py: rank {rank}: device/collective error after the failed staged capture: {e2!r}',
                              file=sys.stderr, flush=True)
                        sys.exit(3)
            elif world == 1 and not args.split_graph:
                g_step = GraphedCallable(whole_step)                 # forward + ELBO + backward + Adam: one graph
            else:
                g_fb = GraphedCallable(fwd_bwd)                      # all-reduce stays an eager RCCL call
                g_adam = GraphedCallable(adam_step, warmup=1)
            # same starting point for every N: initial parameters, zero Adam moments, step counter 0
            with torch.no_grad():
                opt.bucket.flat_p.copy_(p0)
                opt.exp_avg.zero_(); opt.exp_avg_sq.zero_()
                opt.steps = 0
                if opt.step_dev is not None:
                    opt.step_dev.zero_()

        def step(k):
            # minibatch -> static buffers: one multi-tensor copy kernel (two hipMemcpyAsync blits cost 10 us each)
            torch._foreach_copy_([x_in, y_in], [xs[k % n_batches], ys[k % n_batches]])
            if staged and use_graph:
                loss = g_seq()[0]
                dp.check_drained()               # (adam_step's own check only runs at capture time)
                g_adam()
            elif staged:
                for gi in range(groups['n']):
                    out = stage_group_fn(gi)()
                    loss = out if gi == 0 else loss
                    exchange_fn(gi)()
                adam_step()
            elif not use_graph:
                loss = fwd_bwd()
                dp.allreduce_grads(gather=False)
                adam_step()
            elif world == 1 and not args.split_graph:
                loss = g_step()
            else:
                loss = g_fb()
                dp.allreduce_grads(gather=False)
                g_adam()
            return loss

        for k in range(args.warmup):
            step(k)
        barrier()
        t0 = time.perf_counter()
        for k in range(args.steps):
            loss = step(args.warmup + k)
        barrier()
        elapsed = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([elapsed], device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        loss_total = loss.detach().clone().reshape(1)
        if world > 1:                                   # the ranks' objectives SUM to the single-GPU loss (nsgp/dist.py)
            dist.all_reduce(loss_total, op=dist.ReduceOp.SUM)
        final_loss = float(loss_total.item())

        # roofline of the dominant kernel family: re-run the same steps EAGERLY with HIP events around
        # every GEMM launch (kept out of the timed region above)
        timer = GemmTimer()
        ops.set_gemm_timer(timer)
        nprof = min(args.steps, 5)
        for k in range(nprof):
            x_in.copy_(xs[k % n_batches]); y_in.copy_(ys[k % n_batches])
            fwd_bwd()
            dp.allreduce_grads(gather=False)
            adam_step()
        gemm_ms, gemm_flops, gemm_launches = timer.summary(torch.float32)
        g64_ms, g64_flops, g64_launches = timer.summary(torch.float64)
        acc_ms, acc_flops, acc_launches = timer.summary('f64acc')
        i8_ms, i8_flops, i8_launches = timer.summary('i8')
        ops.set_gemm_timer(None)

    result = None
    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        achieved = gemm_flops / (gemm_ms * 1e-3) / 1e12
        sus = mfma_sustained(device)
        result = {
            # a "step" is one fwd+ELBO+bwd+Adam pass over a 4096-row minibatch.  Weak scaling: an N-GPU iteration is N
            # such passes (one per rank, global batch 4096 N) joined by one RCCL all-reduce, so value = N x iterations/s;
            # strong scaling: the N ranks share one 4096-row minibatch, value = iterations/s.
            'metric': 'dsvi_elbo_steps_per_sec',
            'value': round((world if weak else 1) * args.steps / elapsed, 3), 'unit': 'steps/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(ms_per_step, 3),
            'higher_is_better': True, 'scaling': ('weak' if weak else 'strong'), 'vs_baseline': None, 'dtype': 'f32',
            'data': 'synthetic',
            'config': {'workload': '2-layer DSVI DeepGP (hidden 3->2 + last 2->1), M=1024, S=10, minibatch 4096 '
                                   + ('per GPU ' if weak else '(global, split over the GPUs) ')
                                   + 'of synthetic N=1e5 spatio-temporal grid; fwd+ELBO+bwd+Adam',
                       'M': M_INDUCING, 'S': S_SAMPLES, 'per_gpu_batch': hi - lo,
                       'global_batch': gbatch, 'N': N_DATA, 'parallelism': f'dp{world}',
                       'kzz_cholesky_dtype': 'f64', 'hipgraph': bool(use_graph)},
            'iterations_per_sec': round(args.steps / elapsed, 3),
            'rows_per_sec': round(gbatch * args.steps / elapsed, 1),
            **({'analysis_split_graph': 'two graph replays per step, as in an N>1 run'} if args.split_graph else {}),
            **({'analysis_fuse_kzx': 'Kzx generated inside the forward projection (never materialised in the forward pass)'}
               if args.fuse_kzx else {}),
            **({'analysis_fp32_whiten': 'fp32 whitening product (round-1 arithmetic): NOT the shipped precision'}
               if args.fp32_whiten else {}),
            **({'analysis_rehearse_rccl': 'one-rank RCCL group, every collective of the N>1 step issued'} if rehearse else {}),
            **({'gradient_exchange': {
                'mode': 'staged backward: per-stage sum-all-reduce overlapped with the following stage',
                'backward_stages': groups['stages'], 'exchange_groups': groups['n'],
                **({'hipgraph_capture_failed_ran_eagerly': staged_graph_error} if staged_graph_error else {}),
                'group_bytes': [4 * (opt.bucket.segments[g][1] - opt.bucket.segments[g][0]) if g in opt.bucket.segments else 0
                                for g in range(groups['n'])]}} if staged else
               ({'gradient_exchange': {'mode': 'one sum-all-reduce of the flat bucket after the backward pass',
                                       'group_bytes': [4 * opt.bucket.numel]}} if world > 1 else {})),
            **({'analysis': f'one rank\'s share of a {share}-rank job (rows [0, {hi - lo}) of each minibatch), '
                            'no collective; NOT a --gpus result'} if share > 1 else {}),
            'final_loss': round(final_loss, 5),
            'roofline': {'bound': 'mfma', 'kernel': 'gemm_kernel<float,128,128,*,*> (all f32 GEMM launches of a step)',
                         'achieved': round(achieved, 2), 'peak': MFMA_F32_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                         'frac': round(achieved / MFMA_F32_PEAK_TFLOPS, 4), **gemm_traffic(world, share),
                         # context, not the contract's `peak`: the rate a register-only MFMA loop sustains on this chip, and
                         # the clock it holds meanwhile (the 157.3 above assumes 2.4 GHz)
                         'sustained_mfma_measured': sus['f32']['rate'], 'clock_GHz_under_mfma_load': sus['f32']['clock_GHz'],
                         'frac_of_sustained': round(achieved / sus['f32']['rate'], 4),
                         'gemm_ms_per_step': round(gemm_ms / nprof, 3),
                         'gemm_launches_per_step': gemm_launches // nprof,
                         'algorithmic_gflop_per_step': round(gemm_flops / nprof / 1e9, 2),
                         'f64_gemm_ms_per_step': round(g64_ms / nprof, 3),
                         'f64_gemm_launches_per_step': g64_launches // nprof},
            # the whitened projection A = L^-1 Kzx, accumulated in float64 on the float32 Kzx (settings.whiten_matmul_f64):
            # float64 MFMA, priced against the 78.6 TFLOP/s float64 matrix peak
            'f64acc_projection': ({'ms_per_step': round(acc_ms / nprof, 3), 'launches_per_step': acc_launches // nprof,
                                   'algorithmic_gflop_per_step': round(acc_flops / nprof / 1e9, 2),
                                   'achieved': round(acc_flops / (acc_ms * 1e-3) / 1e12, 2), 'peak': MFMA_F64_PEAK_TFLOPS,
                                   'unit': 'TFLOP/s', 'frac': round(acc_flops / (acc_ms * 1e-3) / 1e12 / MFMA_F64_PEAK_TFLOPS, 4),
                                   'sustained_mfma_measured': sus['f64']['rate'],
                                   'frac_of_sustained': round(acc_flops / (acc_ms * 1e-3) / 1e12 / sus['f64']['rate'], 4)}
                                  if acc_launches else None),
            # the same projection on the int8 matrix cores (settings.whiten_matmul_i8, default): exact digit-plane products.
            # `achieved` counts the int8 operations actually issued (14 plane products per multiply-add of the float64
            # product it replaces; 15 for layers with five Kzx planes are counted as 14) against the 5 POP/s dense int8 peak;
            # `f64_equivalent_TFLOPs` is the float64 product's flops over the same time (the float64 MFMA peak is 78.6)
            'i8_projection': ({'ms_per_step': round(i8_ms / nprof, 3), 'launches_per_step': i8_launches // nprof,
                               'algorithmic_gflop_per_step': round(i8_flops / nprof / 1e9, 2),
                               'achieved': round(I8_PLANE_PRODUCTS * i8_flops / (i8_ms * 1e-3) / 1e12, 1), 'peak': MFMA_I8_PEAK_TOPS,
                               'unit': 'TOP/s (int8)', 'frac': round(I8_PLANE_PRODUCTS * i8_flops / (i8_ms * 1e-3) / 1e12 / MFMA_I8_PEAK_TOPS, 4),
                               'f64_equivalent_TFLOPs': round(i8_flops / (i8_ms * 1e-3) / 1e12, 2),
                               'sustained_mfma_measured': sus['i8']['rate'],
                               'frac_of_sustained': round(I8_PLANE_PRODUCTS * i8_flops / (i8_ms * 1e-3) / 1e12 / sus['i8']['rate'], 4)}
                              if i8_launches else None),
        }
        if world == 1:
            if not args.no_build_chol:
                result.update(gibbs_chol_ms(device))
            if not args.no_cpu_baseline:
                torch.set_num_threads(host_cores())
                v, nsteps = cpu_baseline(x_all, y_all, idx, mirror=True)
                v2, nsteps2 = cpu_baseline(x_all, y_all, idx, mirror=False)
                result['cpu_baseline'] = {'value': round(v, 4), 'unit': 'steps/s', 'cores': torch.get_num_threads(),
                                          'kind': 'port',
                                          'sample': f'{nsteps} full DSVI steps (first discarded) of the same '
                                                    'M=1024/S=10/B=4096 workload, oracle in gpytorch-mirror mode',
                                          # the same arithmetic WITHOUT the [recalled] per-sample Kzz / Cholesky recomputation
                                          'value_no_mirror': round(v2, 4),
                                          'sample_no_mirror': f'{nsteps2} steps, Kzz + Cholesky once per layer'}
                result['speedup_vs_cpu'] = round(result['value'] / v, 1)
                result['speedup_vs_cpu_no_mirror'] = round(result['value'] / v2, 1)
        print(json.dumps(result), flush=True)
    if world > 1 or rehearse:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
