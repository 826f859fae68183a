"""Device ops: thin, checked wrappers over the C ABI (include/nsgp.h) + torch.autograd.Functions.

Every function here requires CUDA (ROCm) tensors and runs a hand-written gfx950 kernel on torch's
current stream; torch only owns memory, streams and the autograd graph edges.  There is no CPU or
eager fallback: a CPU tensor or a missing library raises `BackendError`.

Raw ops (no autograd) are lower-case functions returning new tensors; autograd entry points are the
`*Fn` classes and the lower-case convenience wrappers at the bottom (`gibbs_kernel`, `rbf_kernel`,
`ps2d_kernel`, `matmul`, `chol_inv`, ...).
"""
import ctypes

import torch

from . import _lib
from ._lib import BackendError

GEMM_A_LOWER, GEMM_A_UPPER, GEMM_B_LOWER, GEMM_B_UPPER, GEMM_C_LOWER, GEMM_NO_SPLITK, GEMM_C_NOFILL = 1, 2, 4, 8, 16, 32, 64
GEMM_C_HALFDIAG = 128


# --------------------------------------------------------------------------------------------
# plumbing
# --------------------------------------------------------------------------------------------
def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _sfx(t):
    if t.dtype == torch.float32:
        return 'f32'
    if t.dtype == torch.float64:
        return 'f64'
    raise BackendError(f'nsgp kernels compute in float32/float64, got {t.dtype}')


def _chk(*ts):
    """All tensors on the same CUDA device, same floating dtype."""
    first = None
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise BackendError('nsgp ops need CUDA (ROCm) tensors: the HIP backend is the only backend '
                               '(no CPU fallback).  Move the model/data to the GPU.')
        if first is None:
            first = t
        elif t.dtype != first.dtype or t.device != first.device:
            raise BackendError(f'mixed dtype/device: {t.dtype}/{t.device} vs {first.dtype}/{first.device}')
    _sfx(first)
    if first.device.index != torch.cuda.current_device():
        # every wrapper launches on torch.cuda.current_stream() of the CURRENT device: device-1 pointers on device 0's
        # stream would be a GPU memory fault, not an error code
        raise BackendError(f'tensors live on {first.device} but the current device is cuda:{torch.cuda.current_device()}: '
                           'call torch.cuda.set_device(...) (one process per GPU) or wrap the call in torch.cuda.device(...)')
    return first


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


def _ws(nbytes, device):
    return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=device)


# --------------------------------------------------------------------------------------------
# Gradient sinks: where a large parameter's gradient should be WRITTEN (the flat gradient bucket of nsgp.optim.FlatBucket)
# instead of being handed to autograd in a fresh buffer and copied there afterwards.  A backward kernel that produces the
# gradient of a registered parameter asks `grad_sink(t)` (t: the parameter as the node received it -- same storage):
#   * (view, False): nothing is in place yet -> write the gradient into `view` and RETURN `view` as the gradient (autograd's
#     AccumulateGrad adopts it as p.grad: the bucket is filled without a copy);
#   * (view, True):  the view already holds a term of this gradient (the KL term, an earlier application of a tied layer,
#     or -- gradient accumulation over several backward passes -- p.grad itself) -> ACCUMULATE into `view` and return
#     None for that input.
# "Already holds a term" cannot be read off p.grad inside one backward pass (the engine sums the incoming gradients of a leaf
# in its input buffer and sets p.grad only after the LAST of them), so a sink remembers the autograd graph task that wrote
# it first; across passes p.grad decides, so any way of clearing gradients is safe.
# --------------------------------------------------------------------------------------------
_grad_sinks = {}
_grad_sinks_on = True


class grad_sinks:
    """Context manager / switch: `with ops.grad_sinks(False): ...` makes every backward pass inside hand its gradients
    over in fresh buffers, as plain torch does.  Needed when gradients are taken with `torch.autograd.grad` (or kept across
    `zero_grad()`): a sink's gradient IS a view of the optimiser's flat gradient bucket, so the next backward pass would
    overwrite a tensor the caller still holds.  (Inside an ordinary `loss.backward()` the view becomes `p.grad` and the
    aliasing is the point: the bucket is filled without a copy.)"""

    def __init__(self, enabled):
        self.enabled, self.prev = bool(enabled), None

    def __enter__(self):
        global _grad_sinks_on
        self.prev, _grad_sinks_on = _grad_sinks_on, self.enabled
        return self

    def __exit__(self, *exc):
        global _grad_sinks_on
        _grad_sinks_on = self.prev
        return False


class _GradSink:
    def __init__(self, param, flat_g, off):
        import weakref
        self.param = weakref.ref(param)
        self.flat_g, self.off, self.numel, self.shape = flat_g, int(off), param.numel(), tuple(param.shape)
        self.task = None                                  # id of the backward pass (graph task) that wrote first

    def view(self):
        return self.flat_g[self.off:self.off + self.numel].view(self.shape)


def register_grad_sink(param, flat_g, off):
    _grad_sinks[param.data_ptr()] = _GradSink(param, flat_g, off)


def unregister_grad_sinks(flat_g):
    for k in [k for k, v in _grad_sinks.items() if v.flat_g is flat_g]:
        del _grad_sinks[k]


def grad_sink(t):
    """(view shaped like t, accumulate) for a tensor that IS a registered parameter (same storage, same number of
    elements, contiguous) inside a backward pass, else None."""
    s = _grad_sinks.get(t.data_ptr()) if _grad_sinks_on else None
    if s is None:
        return None
    task = torch._C._current_graph_task_id()
    p = s.param()
    if task < 0 or p is None or p.data_ptr() != t.data_ptr() or t.numel() != s.numel or not t.is_contiguous() \
            or t.dtype != s.flat_g.dtype or t.device != s.flat_g.device:
        return None
    v = s.view()
    if p.grad is not None:
        if p.grad.data_ptr() != v.data_ptr():
            return None                               # something else owns p.grad: the ordinary path accumulates into it
        s.task = task
        return v.view(t.shape), True                  # an earlier pass (or stage) left its gradient in the view
    if s.task == task:
        return v.view(t.shape), True                  # an earlier node of THIS pass wrote the view
    s.task = task
    return v.view(t.shape), False


def _scalar_dev(v, like):
    """Host number or tensor -> 1-element device tensor of like's dtype (no sync)."""
    if torch.is_tensor(v):
        return v.detach().reshape(1).to(device=like.device, dtype=like.dtype)
    return torch.full((1,), float(v), dtype=like.dtype, device=like.device)


# --------------------------------------------------------------------------------------------
# K1 Gibbs
# --------------------------------------------------------------------------------------------
def _gibbs_args(x1, x2, ell1, ell2):
    ref = _chk(x1, x2, ell1, ell2)
    if x1.dim() != 2 or x2.dim() != 2 or ell1.dim() != 2 or ell2.dim() != 2:
        raise BackendError('gibbs_build: x:(n,D), ell:(D,n) expected')
    n1, D = x1.shape
    n2 = x2.shape[0]
    if x2.shape[1] != D or ell1.shape != (D, n1) or ell2.shape != (D, n2):
        raise BackendError(f'gibbs_build: shape mismatch x1{tuple(x1.shape)} x2{tuple(x2.shape)} '
                           f'ell1{tuple(ell1.shape)} ell2{tuple(ell2.shape)}')
    return ref, n1, n2, D


def _out_matrix(out, shape, ref):
    """Validate a caller-provided contiguous output buffer (or allocate one)."""
    if out is None:
        return torch.empty(shape, dtype=ref.dtype, device=ref.device)
    if tuple(out.shape) != tuple(shape) or out.dtype != ref.dtype or out.device != ref.device \
            or not out.is_contiguous():
        raise BackendError(f'out: contiguous {tuple(shape)} {ref.dtype} tensor on {ref.device} expected')
    return out


def gibbs_build(x1, x2, ell1, ell2, outputscale=None, diag_add=None, out=None):
    """K = os * Gibbs(x1,x2; ell1,ell2) (+ diag_add on the diagonal).  models/gibbs_kernels.py:154-162."""
    ref, n1, n2, D = _gibbs_args(x1, x2, ell1, ell2)
    x1, x2, ell1, ell2 = _c(x1), _c(x2), _c(ell1), _c(ell2)
    os_ = None if outputscale is None else _scalar_dev(outputscale, ref)
    da = None if diag_add is None else _scalar_dev(diag_add, ref)
    K = _out_matrix(out, (n1, n2), ref)
    _lib.call(f'nsgp_gibbs_build_fwd_{_sfx(ref)}', _p(x1), _p(x2), _p(ell1), _p(ell2), n1, n2, D,
              _p(os_), _p(da), _p(K), n2, _stream())
    return K


def gibbs_build_bwd(x1, x2, ell1, ell2, outputscale, G, need_x=False, need_os=True):
    ref, n1, n2, D = _gibbs_args(x1, x2, ell1, ell2)
    _chk(ref, G)
    if G.shape != (n1, n2):
        raise BackendError('gibbs_build_bwd: G shape')
    x1, x2, ell1, ell2, G = _c(x1), _c(x2), _c(ell1), _c(ell2), _c(G)
    os_ = None if outputscale is None else _scalar_dev(outputscale, ref)
    g_l1, g_l2 = torch.empty_like(ell1), torch.empty_like(ell2)
    g_x1 = torch.empty_like(x1) if need_x else None
    g_x2 = torch.empty_like(x2) if need_x else None
    g_os = torch.empty(1, dtype=ref.dtype, device=ref.device) if need_os else None
    lib = _lib.load()
    wsb = lib.nsgp_gibbs_build_bwd_workspace(n1, n2, D, ref.element_size())
    ws = _ws(wsb, ref.device)
    _lib.call(f'nsgp_gibbs_build_bwd_{_sfx(ref)}', _p(x1), _p(x2), _p(ell1), _p(ell2), n1, n2, D, _p(os_),
              _p(G), n2, _p(g_l1), _p(g_l2), _p(g_x1), _p(g_x2), _p(g_os), _p(ws), ws.numel(), _stream())
    return g_l1, g_l2, g_x1, g_x2, g_os


# --------------------------------------------------------------------------------------------
# K2 RBF-ARD (batched)
# --------------------------------------------------------------------------------------------
def _rbf_args(x1, x2, ls, os_):
    ref = _chk(x1, x2, ls, os_)
    if ls.dim() == 1:
        ls = ls.unsqueeze(0)
    os_ = os_.reshape(-1)
    batch, D = ls.shape
    if os_.shape[0] != batch:
        raise BackendError('rbf_build: os must be (batch,)')

    def prep(x):
        if x.dim() == 2:
            if x.shape[1] != D:
                raise BackendError('rbf_build: x last dim != D')
            return _c(x), x.shape[0], 0
        if x.dim() == 3 and x.shape[0] == batch and x.shape[2] == D:
            x = _c(x)
            return x, x.shape[1], x.shape[1] * D
        raise BackendError(f'rbf_build: x shape {tuple(x.shape)} vs batch {batch}, D {D}')
    x1, n1, sx1 = prep(x1)
    x2, n2, sx2 = prep(x2)
    return ref, x1, x2, _c(ls), _c(os_), batch, n1, n2, D, sx1, sx2


def rbf_build(x1, x2, ls, os_, diag_add=0.0, out=None):
    """K[b] = os[b] * exp(-0.5 |(x1-x2)/ls[b]|^2) (+diag_add I).  x:(n,D) shared or (batch,n,D)."""
    ref, x1, x2, ls, os_, batch, n1, n2, D, sx1, sx2 = _rbf_args(x1, x2, ls, os_)
    K = _out_matrix(out, (batch, n1, n2), ref)
    _lib.call(f'nsgp_rbf_build_fwd_{_sfx(ref)}', _p(x1), _p(x2), _p(ls), _p(os_), batch, n1, n2, D, sx1, sx2,
              float(diag_add), _p(K), n2, n1 * n2, _stream())
    return K


def rbf_build_bwd(x1, x2, ls, os_, G, need_x1=True, need_x2=True, sym=False):
    """Returns g_x1:(batch,n1,D) g_x2:(batch,n2,D) (per-batch, caller sums if x was shared) g_ls, g_os.
    sym=True (x1 is x2, the Kzz case): ONE buffer receives the sum of the row- and column-side gradients and is
    returned for both."""
    ref, x1, x2, ls, os_, batch, n1, n2, D, sx1, sx2 = _rbf_args(x1, x2, ls, os_)
    _chk(ref, G)
    G = _c(G).reshape(batch, n1, n2)
    g_x1 = torch.empty((batch, n1, D), dtype=ref.dtype, device=ref.device) if need_x1 else None
    g_x2 = torch.empty((batch, n2, D), dtype=ref.dtype, device=ref.device) if need_x2 else None
    if sym:
        if n1 != n2 or sx1 != sx2 or not (need_x1 and need_x2):
            raise BackendError('rbf_build_bwd: sym needs x1 and x2 of one shape and both gradients')
        g_x2 = g_x1
    g_ls = torch.empty((batch, D), dtype=ref.dtype, device=ref.device)
    g_os = torch.empty((batch,), dtype=ref.dtype, device=ref.device)
    lib = _lib.load()
    ws = _ws(lib.nsgp_rbf_build_bwd_workspace(batch, n1, n2, D, ref.element_size()), ref.device)
    _lib.call(f'nsgp_rbf_build_bwd_{_sfx(ref)}', _p(x1), _p(x2), _p(ls), _p(os_), batch, n1, n2, D, sx1, sx2,
              _p(G), n2, n1 * n2, _p(g_x1), _p(g_x2), _p(g_ls), _p(g_os), _p(ws), ws.numel(), _stream())
    return g_x1, g_x2, g_ls, g_os


# --------------------------------------------------------------------------------------------
# K2' RBF-ARD x Periodic (one launch)
# --------------------------------------------------------------------------------------------
def _rbfper_args(x1, x2, ls_rbf, ls_per, period, os_):
    ref = _chk(x1, x2, ls_rbf, ls_per, period, os_)
    ls_per, period = _c(ls_per.reshape(-1)), _c(period.reshape(-1))
    batch = ls_per.shape[0]
    if period.shape[0] != batch:
        raise BackendError('rbf_periodic_build: ls_per and period must be (batch,)')
    D = x1.shape[-1]
    if ls_rbf is not None:
        ls_rbf = _c(ls_rbf.reshape(batch, -1))
        if ls_rbf.shape[1] != D:
            raise BackendError('rbf_periodic_build: ls_rbf must be (batch, D)')
    if os_ is not None:
        os_ = _c(os_.reshape(-1))
        if os_.shape[0] != batch:
            raise BackendError('rbf_periodic_build: os must be (batch,)')

    def prep(x):
        if x.dim() == 2 and x.shape[1] == D:
            return _c(x), x.shape[0], 0
        if x.dim() == 3 and x.shape[0] == batch and x.shape[2] == D:
            x = _c(x)
            return x, x.shape[1], x.shape[1] * D
        raise BackendError(f'rbf_periodic_build: x shape {tuple(x.shape)} vs batch {batch}, D {D}')
    x1, n1, sx1 = prep(x1)
    x2, n2, sx2 = prep(x2)
    return ref, x1, x2, ls_rbf, ls_per, period, os_, batch, n1, n2, D, sx1, sx2


def rbf_periodic_build(x1, x2, ls_rbf, ls_per, period, os_, diag_add=0.0):
    """K[b] = os[b] RBF-ARD(x; ls_rbf[b]) Periodic(x; ls_per[b], period[b]); ls_rbf / os_ may be None."""
    ref, x1, x2, ls_rbf, ls_per, period, os_, batch, n1, n2, D, sx1, sx2 = _rbfper_args(x1, x2, ls_rbf, ls_per,
                                                                                       period, os_)
    K = torch.empty((batch, n1, n2), dtype=ref.dtype, device=ref.device)
    _lib.call(f'nsgp_rbf_periodic_build_fwd_{_sfx(ref)}', _p(x1), _p(x2), _p(ls_rbf), _p(ls_per), _p(period), _p(os_),
              batch, n1, n2, D, sx1, sx2, float(diag_add), _p(K), n2, n1 * n2, _stream())
    return K


def rbf_periodic_build_bwd(x1, x2, ls_rbf, ls_per, period, os_, G, need_x1=True, need_x2=True):
    ref, x1, x2, ls_rbf, ls_per, period, os_, batch, n1, n2, D, sx1, sx2 = _rbfper_args(x1, x2, ls_rbf, ls_per,
                                                                                       period, os_)
    _chk(ref, G)
    G = _c(G).reshape(batch, n1, n2)
    new = lambda *shp: torch.empty(shp, dtype=ref.dtype, device=ref.device)
    g_x1 = new(batch, n1, D) if need_x1 else None
    g_x2 = new(batch, n2, D) if need_x2 else None
    g_lr = new(batch, D) if ls_rbf is not None else None
    g_lp, g_pe = new(batch), new(batch)
    g_os = new(batch)
    lib = _lib.load()
    ws = _ws(lib.nsgp_rbf_periodic_build_bwd_workspace(batch, n1, n2, D, ref.element_size()), ref.device)
    _lib.call(f'nsgp_rbf_periodic_build_bwd_{_sfx(ref)}', _p(x1), _p(x2), _p(ls_rbf), _p(ls_per), _p(period), _p(os_),
              batch, n1, n2, D, sx1, sx2, _p(G), n2, n1 * n2, _p(g_x1), _p(g_x2), _p(g_lr), _p(g_lp), _p(g_pe), _p(g_os),
              _p(ws), ws.numel(), _stream())
    return g_x1, g_x2, g_lr, g_lp, g_pe, g_os


# --------------------------------------------------------------------------------------------
# K3 Paciorek-Schervish (D = 2)
# --------------------------------------------------------------------------------------------
def _ps_args(x1, x2, s1, s2):
    ref = _chk(x1, x2, s1, s2)
    n1, n2 = x1.shape[0], x2.shape[0]
    if x1.shape != (n1, 2) or x2.shape != (n2, 2) or s1.shape != (n1, 2, 2) or s2.shape != (n2, 2, 2):
        raise BackendError('ps2d_build: x:(n,2), sigma:(n,2,2) expected')
    return ref, n1, n2


def ps2d_build(x1, x2, s1, s2, jitter=1e-5):
    ref, n1, n2 = _ps_args(x1, x2, s1, s2)
    x1, x2, s1, s2 = _c(x1), _c(x2), _c(s1), _c(s2)
    K = torch.empty((n1, n2), dtype=ref.dtype, device=ref.device)
    _lib.call(f'nsgp_ps2d_build_fwd_{_sfx(ref)}', _p(x1), _p(x2), _p(s1), _p(s2), n1, n2, float(jitter), _p(K), n2,
              _stream())
    return K


def ps2d_build_bwd(x1, x2, s1, s2, jitter, G):
    ref, n1, n2 = _ps_args(x1, x2, s1, s2)
    _chk(ref, G)
    x1, x2, s1, s2, G = _c(x1), _c(x2), _c(s1), _c(s2), _c(G)
    g1, g2 = torch.empty_like(s1), torch.empty_like(s2)
    lib = _lib.load()
    ws = _ws(lib.nsgp_ps2d_build_bwd_workspace(n1, n2, ref.element_size()), ref.device)
    _lib.call(f'nsgp_ps2d_build_bwd_{_sfx(ref)}', _p(x1), _p(x2), _p(s1), _p(s2), n1, n2, float(jitter), _p(G), n2,
              _p(g1), _p(g2), _p(ws), ws.numel(), _stream())
    return g1, g2


# --------------------------------------------------------------------------------------------
# MFMA GEMM
# --------------------------------------------------------------------------------------------
def _mat_view(t, trans):
    """Return (tensor, rows, cols, row_stride, col_stride, batch, batch_stride) of op(t) without copying
    when one of the two trailing strides is 1; otherwise make it contiguous."""
    if t.dim() not in (2, 3):
        raise BackendError('gemm: operands must be 2-D or 3-D (batched)')
    sr, sc = t.stride(-2), t.stride(-1)
    ok = (sr == 1 or sc == 1) and (t.dim() == 2 or t.shape[0] == 1 or t.stride(0) >= 0)
    if t.shape[-1] == 1 and sr != 1:
        sc = 1
    if t.shape[-2] == 1 and sc != 1:
        sr = 1
    if not ok or not (sr == 1 or sc == 1):
        t = t.contiguous()
        sr, sc = t.stride(-2), t.stride(-1)
    r, c = t.shape[-2], t.shape[-1]
    if trans:
        r, c, sr, sc = c, r, sc, sr
    nb = t.shape[0] if t.dim() == 3 else 1
    sb = t.stride(0) if t.dim() == 3 else 0
    return t, r, c, sr, sc, nb, sb


_gemm_timer = None


def set_gemm_timer(timer):
    """bench.py hook: `timer(fn, flops, dtype)` must call fn() (the launch) and may bracket it with HIP events.
    flops = algorithmic 2*M*N*K per batch element, halved for a triangular operand / lower-only output."""
    global _gemm_timer
    _gemm_timer = timer


def gemm(A, B, ta=False, tb=False, alpha=1.0, beta=0.0, out=None, flags=0):
    """out = alpha * op(A) @ op(B) + beta * out on the matrix cores.  2-D or batched 3-D operands
    (a 2-D operand broadcasts against a 3-D one)."""
    ref = _chk(A, B, out)
    A, M, K, sam, sak, nba, sba = _mat_view(A, ta)
    B, K2, N, sbk, sbn, nbb, sbb = _mat_view(B, tb)
    if K != K2:
        raise BackendError(f'gemm: inner dims {K} vs {K2}')
    nb = max(nba, nbb)
    if nba not in (1, nb) or nbb not in (1, nb):
        raise BackendError('gemm: batch mismatch')
    if nba == 1:
        sba = 0
    if nbb == 1:
        sbb = 0
    batched = A.dim() == 3 or B.dim() == 3
    shape = (nb, M, N) if batched else (M, N)
    if out is None:
        if beta != 0.0:
            raise BackendError('gemm: beta != 0 needs out')
        out = torch.empty(shape, dtype=ref.dtype, device=ref.device)
    else:
        if tuple(out.shape) != shape or not out.is_contiguous():
            raise BackendError(f'gemm: out must be contiguous {shape}, got {tuple(out.shape)}')
    lib = _lib.load()
    wsb = 0 if (flags & GEMM_NO_SPLITK) else lib.nsgp_gemm_workspace(M, N, K, nb, 1, ref.element_size(), int(flags))
    ws = _ws(wsb, ref.device) if wsb else None
    def launch():
        _lib.call(f'nsgp_gemm_{_sfx(ref)}', M, N, K, float(alpha), _p(A), sam, sak, sba, 0, _p(B), sbk, sbn, sbb, 0,
                  float(beta), _p(out), N, M * N, 0, nb, 1, int(flags), _p(ws), ws.numel() if ws is not None else 0,
                  _stream())
    if _gemm_timer is not None:
        tri = flags & (GEMM_A_LOWER | GEMM_A_UPPER | GEMM_B_LOWER | GEMM_B_UPPER | GEMM_C_LOWER)
        _gemm_timer(launch, 2.0 * M * N * K * nb * (0.5 if tri else 1.0), ref.dtype)
    else:
        launch()
    return out


# --------------------------------------------------------------------------------------------
# K4 / K5 Cholesky, triangular inverse
# --------------------------------------------------------------------------------------------
def potrf(A, check=False, overwrite=False):
    """Lower Cholesky factor of a (batched) SPD matrix; returns (L, info) with info an int32 device
    tensor (LAPACK convention).  `check=True` syncs and raises on failure; `overwrite=True` factors a
    contiguous input in place."""
    ref = _chk(A)
    if A.dim() not in (2, 3) or A.shape[-1] != A.shape[-2]:
        raise BackendError('potrf: square (batched) matrix expected')
    L = A if (overwrite and A.is_contiguous()) else A.contiguous().clone()
    n = L.shape[-1]
    batch = L.shape[0] if L.dim() == 3 else 1
    info = (torch.empty if n > 0 else torch.zeros)(batch, dtype=torch.int32, device=ref.device)   # set by the first panel
    lib = _lib.load()
    ws = _ws(lib.nsgp_potrf_workspace(n, batch, ref.element_size()), ref.device)
    _lib.call(f'nsgp_potrf_{_sfx(ref)}', _p(L), n, n, n * n, batch, _p(info), _p(ws), ws.numel(), _stream())
    if check:
        bad = int(info.max().item())
        if bad:
            raise BackendError(f'potrf: leading minor {bad} is not positive definite')
    return L, info


def trtri(L):
    """X = L^-1 for a (batched) lower-triangular L (strict upper of L ignored, of X zero)."""
    ref = _chk(L)
    if L.dim() not in (2, 3) or L.shape[-1] != L.shape[-2]:
        raise BackendError('trtri: square (batched) matrix expected')
    L = _c(L)
    n = L.shape[-1]
    batch = L.shape[0] if L.dim() == 3 else 1
    X = torch.empty_like(L)
    lib = _lib.load()
    ws = _ws(lib.nsgp_trtri_workspace(n, batch, ref.element_size()), ref.device)
    _lib.call(f'nsgp_trtri_{_sfx(ref)}', _p(L), n, n, n * n, _p(X), n, n * n, batch, _p(ws), ws.numel(), _stream())
    return X


def potrf_trtri_(A, want_f32=False):
    """(X, info) with X = chol(A)^-1 (lower triangular) for a contiguous (batched) SPD matrix that is CONSUMED: on return A
    holds intermediate data, not a factor.  One chain of launches without the factor's write-back pass.
    want_f32=True (float64 A): returns (X, info, X32) with a float32 copy of X -- written by the same launches when the
    in-launch inverse runs, by a cast otherwise."""
    ref = _chk(A)
    if A.dim() not in (2, 3) or A.shape[-1] != A.shape[-2] or not A.is_contiguous():
        raise BackendError('potrf_trtri_: contiguous square (batched) matrix expected')
    n = A.shape[-1]
    batch = A.shape[0] if A.dim() == 3 else 1
    X = torch.empty_like(A)
    info = (torch.empty if n > 0 else torch.zeros)(batch, dtype=torch.int32, device=ref.device)
    lib = _lib.load()
    ws = _ws(lib.nsgp_potrf_workspace(n, batch, ref.element_size()) + lib.nsgp_trtri_workspace(n, batch, ref.element_size()),
             ref.device)
    if want_f32 and ref.dtype == torch.float64:
        import ctypes
        X32 = torch.empty(A.shape, dtype=torch.float32, device=ref.device)
        wrote = ctypes.c_int(0)
        _lib.call('nsgp_potrf_trtri_f64_w32', _p(A), n, n, n * n, batch, _p(info), _p(X), n, n * n, _p(X32),
                  ctypes.addressof(wrote), _p(ws), ws.numel(), _stream())
        if not wrote.value:
            X32 = cast(X, torch.float32)
        return X, info, X32
    _lib.call(f'nsgp_potrf_trtri_{_sfx(ref)}', _p(A), n, n, n * n, batch, _p(info), _p(X), n, n * n, _p(ws), ws.numel(),
              _stream())
    if want_f32:
        return X, info, (X if ref.dtype == torch.float32 else cast(X, torch.float32))
    return X, info


def chol_bwd_phi_sym(P):
    ref = _chk(P)
    P = _c(P)
    n = P.shape[-1]
    batch = P.shape[0] if P.dim() == 3 else 1
    S = torch.empty_like(P)
    _lib.call(f'nsgp_chol_bwd_phi_sym_{_sfx(ref)}', _p(P), _p(S), n, n, n * n, batch, _stream())
    return S


def scale_diag_(P, factor):
    """P[..., i, i] *= factor in place (batch of square matrices)."""
    ref = _chk(P)
    if not P.is_contiguous():
        raise BackendError('scale_diag_: contiguous matrices expected')
    n = P.shape[-1]
    batch = P.shape[0] if P.dim() == 3 else 1
    _lib.call(f'nsgp_scale_diag_{_sfx(ref)}', _p(P), n, n, n * n, batch, float(factor), _stream())
    return P


def cast(t, dtype, out=None):
    """float32 <-> float64 copy on the current stream (row-major, any leading shape).  `out`: a contiguous tensor of the
    target dtype and the same number of elements to write into (e.g. a slice of a batched buffer) instead of a new one."""
    _chk(t)
    if t.dtype == dtype and out is None:
        return t
    t = _c(t)
    if out is None:
        out = torch.empty(t.shape, dtype=dtype, device=t.device)
    elif out.dtype != dtype or out.numel() != t.numel() or not out.is_contiguous() or out.device != t.device or t.dtype == dtype:
        raise BackendError('cast(out=...): contiguous tensor of the target dtype with as many elements, on the same device')
    cols = t.shape[-1] if t.dim() else 1
    rows = t.numel() // max(cols, 1)
    name = 'nsgp_cast_f64_to_f32' if dtype == torch.float32 else 'nsgp_cast_f32_to_f64'
    _lib.call(name, _p(t), cols, _p(out), cols, rows, cols, _stream())
    return out


# --------------------------------------------------------------------------------------------
# K6 / K7 raw ops
# --------------------------------------------------------------------------------------------
def svgp_colstats(A, C, m, base):
    ref = _chk(A, C, m, base)
    A, C, m, base = _c(A), _c(C), _c(m), _c(base.reshape(-1))
    batch, M, n = A.shape
    if C.shape != A.shape or m.shape != (batch, M) or base.shape != (batch,):
        raise BackendError('svgp_colstats: shapes')
    mean = torch.empty((batch, n), dtype=ref.dtype, device=ref.device)
    var = torch.empty_like(mean)
    _lib.call(f'nsgp_svgp_colstats_{_sfx(ref)}', _p(A), _p(C), _p(m), _p(base), batch, M, n, _p(mean), _p(var),
              _stream())
    return mean, var


def svgp_colstats_bwd(A, C, m, gmean, gvar):
    ref = _chk(A, C, m, gmean, gvar)
    A, C, m, gmean, gvar = _c(A), _c(C), _c(m), _c(gmean), _c(gvar)
    batch, M, n = A.shape
    if gmean.shape != (batch, n) or gvar.shape != (batch, n):
        raise BackendError('svgp_colstats_bwd: shapes')
    Abar, C2 = torch.empty_like(A), torch.empty_like(C)
    mbar = torch.empty_like(m)
    _lib.call(f'nsgp_svgp_colstats_bwd_{_sfx(ref)}', _p(A), _p(C), _p(m), _p(gmean), _p(gvar), batch, M, n,
              _p(Abar), _p(C2), _p(mbar), _stream())
    return Abar, C2, mbar


def _timed(launch, flops, dtype):
    if _gemm_timer is not None:
        _gemm_timer(launch, flops, dtype)
    else:
        launch()


def _affine_args(affine, batch, n, ref):
    """(x, w, c) of an affine prior mean -> pointer / stride arguments of the *_affine entry points.
    x:(n,D) shared by the batch or (batch,n,D); w:(D,) shared or (batch,D) or None; c:(1,) shared or (batch,) or None."""
    x, w, c = affine
    x = _c(x)
    D = x.shape[-1]
    if x.shape[-2] != n or (x.dim() == 3 and x.shape[0] != batch) or x.dim() not in (2, 3):
        raise BackendError('affine prior mean: x shape')
    sxb = n * D if x.dim() == 3 else 0
    swb = scb = 0
    if w is not None:
        w = _c(w.reshape(-1, D))
        if w.shape[0] not in (1, batch):
            raise BackendError('affine prior mean: weights shape')
        swb = D if (w.shape[0] == batch and batch > 1) else 0
    if c is not None:
        c = _c(c.reshape(-1))
        if c.shape[0] not in (1, batch):
            raise BackendError('affine prior mean: constant shape')
        scb = 1 if (c.shape[0] == batch and batch > 1) else 0
    _chk(ref, x, *[t for t in (w, c) if t is not None])
    return x, sxb, D, w, swb, c, scb


def svgp_kzx_fusable(W64f, Z, x, n):
    """True when A = W Kzx can run with Kzx generated inside the GEMM loader (nsgp_svgp_kzx_gemm_colstats_f64acc):
    float32 layer with its float64 W, D <= 4, whole tiles."""
    if W64f is None or W64f.dtype != torch.float64 or Z.dtype != torch.float32 or not W64f.is_contiguous():
        return False
    batch, M, D = Z.shape
    return bool(_lib.load().nsgp_svgp_kzx_gemm_supported(_p(W64f), M, n, batch, D))


def _project_a_i8(W64f, kZ, kx, kls, kos, m, A, part_dot, part_sq, T, p64, planes, flops, kzx_out=None):
    """A = W Kzx as the exact int8 digit-plane product (csrc/gemm_i8.hip): digit planes of W and of Kzx (evaluated in
    float64 from Z, x, ls, os), then the product with its column-statistic partials (T tile rows per batch element).
    kzx_out: a list that receives the float32 Kzx (b, M, n) the plane-build kernel writes along (for the backward pass)."""
    lib = _lib.load()
    batch, M, D = kZ.shape
    n = kx.shape[-2]
    dev, st = A.device, _stream()
    Wd = torch.empty(int(lib.nsgp_i8_w_planes_bytes(batch, M)), dtype=torch.uint8, device=dev)
    Kd = torch.empty(int(lib.nsgp_i8_k_planes_bytes(batch, M, n, planes)), dtype=torch.uint8, device=dev)
    wsc = torch.empty((batch, M), dtype=torch.float64, device=dev)
    ksc = torch.empty((batch,), dtype=torch.float64, device=dev)
    K32 = torch.empty((batch, M, n), dtype=torch.float32, device=dev) if kzx_out is not None else None
    _lib.call('nsgp_i8_slice_w_f64', _p(W64f), batch, M, _p(Wd), _p(wsc), st)
    _lib.call('nsgp_i8_rbf_build_f32', _p(kZ), _p(kx), n * D if kx.dim() == 3 else 0, _p(kls), _p(kos), batch, M, n, D,
              planes, _p(Kd), _p(ksc), _p(K32), st)
    if kzx_out is not None:
        kzx_out.append(K32)
    _timed(lambda: _lib.call('nsgp_svgp_tri_gemm_colstats_i8', _p(Wd), _p(wsc), _p(Kd), _p(ksc), planes, _p(m), batch, M, n,
                             _p(A), _p(part_dot), _p(part_sq), T, 1 if p64 else 0, st), flops, 'i8')


def svgp_project(W, Kzx, Lq, m, base, base_add=0.0, affine=None, W64f=None, kernel_inputs=None, Kzx64=None, Lq64=None,
                 i8_inputs=None, i8_planes=4, i8_kzx_out=None):
    """Fused K6 forward: A = W Kzx, C = Lq^T A (triangular MFMA GEMMs) with the column statistics reduced in
    the GEMM epilogues.  W, Lq:(b,M,M) lower; Kzx:(b,M,n); m:(b,M); base:(b,).
    Returns A, C, mean = A^T m (+ the affine prior mean x w + c, `affine` = (x, w, c)), var = base + base_add +
    colsum(C^2 - A^2).
    W64f: the float64 W of a float32 layer -- A is then accumulated in float64 on the float32 Kzx and rounded once
    (nsgp_svgp_tri_gemm_colstats_f64acc: the reference's float64 solve); W itself (float32) is only used by the backward.
    kernel_inputs=(Z, x, ls, os) with Kzx=None (and W64f, `svgp_kzx_fusable`): Kzx is never materialised, its tiles are
    generated inside the loader of the first product from Z:(b,M,D), x:(n,D) or (b,n,D), ls:(b,D), os:(b,)."""
    i8 = i8_inputs is not None
    if i8:
        # A = W Kzx on the int8 matrix cores (csrc/gemm_i8.hip): exact int32 accumulation of 14 digit-plane products of the
        # float64 W and of Kzx evaluated in float64 from (Z, x, ls, os); Kzx itself is never materialised in the forward pass
        if Kzx is not None or Kzx64 is not None or kernel_inputs is not None or W64f is None:
            raise BackendError('svgp_project: i8_inputs=(Z, x, ls, os) comes with W64f and without Kzx / Kzx64 / kernel_inputs')
        kZ, kx, kls, kos = i8_inputs
        ref = _chk(W, Lq, m, base, kZ, kx, kls, kos)
        kZ, kx, kls, kos = _c(kZ), _c(kx), _c(kls), _c(kos.reshape(-1))
        W, Lq, m, base = _c(W), _c(Lq), _c(m), _c(base.reshape(-1))
        batch, M, D = kZ.shape
        n = kx.shape[-2]
        if ref.dtype != torch.float32 or kx.shape[-1] != D or (kx.dim() == 3 and kx.shape[0] != batch) \
                or kls.shape != (batch, D) or kos.shape != (batch,) or D > 4 or not _lib.load().nsgp_i8_supported(M):
            raise BackendError('svgp_project: i8_inputs shapes (float32 layer, D <= 4, M <= 4096)')
    b64 = Kzx64 is not None
    if b64:
        if Kzx is not None or W64f is None or Kzx64.dtype != torch.float64 or Kzx64.dim() != 3:
            raise BackendError('svgp_project: Kzx64 (float64 (b,M,n), with W64f, instead of Kzx) expected')
        ref = _chk(W, Lq, m, base)
        if Kzx64.device != ref.device:
            raise BackendError('svgp_project: Kzx64 device')
        W, Kzx64, Lq, m, base = _c(W), _c(Kzx64), _c(Lq), _c(m), _c(base.reshape(-1))
        batch, M, n = Kzx64.shape
    fused = Kzx is None and not b64 and not i8
    if fused:
        if kernel_inputs is None or W64f is None:
            raise BackendError('svgp_project: Kzx=None needs kernel_inputs and W64f')
        kZ, kx, kls, kos = kernel_inputs
        ref = _chk(W, Lq, m, base, kZ, kx, kls, kos)
        kZ, kx, kls, kos = _c(kZ), _c(kx), _c(kls), _c(kos.reshape(-1))
        W, Lq, m, base = _c(W), _c(Lq), _c(m), _c(base.reshape(-1))
        batch, M, D = kZ.shape
        n = kx.shape[-2]
        if kx.shape[-1] != D or (kx.dim() == 3 and kx.shape[0] != batch) or kls.shape != (batch, D) or kos.shape != (batch,):
            raise BackendError('svgp_project: kernel_inputs shapes')
    elif not b64 and not i8:
        ref = _chk(W, Kzx, Lq, m, base)
        W, Kzx, Lq, m, base = _c(W), _c(Kzx), _c(Lq), _c(m), _c(base.reshape(-1))
        batch, M, n = Kzx.shape
    if W.shape != (batch, M, M) or Lq.shape != (batch, M, M) or m.shape != (batch, M) or base.shape != (batch,):
        raise BackendError('svgp_project: shapes')
    lib = _lib.load()
    T = int(lib.nsgp_svgp_colstats_tiles(M, n, batch, ref.element_size()))
    A = torch.empty((batch, M, n), dtype=ref.dtype, device=ref.device)
    C = torch.empty_like(A)
    sfx, st = _sfx(ref), _stream()
    flops = 1.0 * M * M * n * batch                      # 2 M M n / 2 (triangular operand)
    if W64f is not None:
        if ref.dtype != torch.float32 or W64f.dtype != torch.float64 or W64f.shape != (batch, M, M) \
                or W64f.device != ref.device:
            raise BackendError('svgp_project: W64f must be the float64 (b,M,M) W of a float32 layer')
        W64f = _c(W64f)
        T64 = int(lib.nsgp_i8_tiles(M)) if i8 else int(lib.nsgp_svgp_f64acc_tiles_for(M, n, batch))   # tile rows of product 1
        if i8 and Lq64 is not None:               # product 2 on the float64-accumulating kernel: ITS tile rows for part[2]
            T64 = max(T64, int(lib.nsgp_svgp_f64acc_tiles_for(M, n, batch)))
        p64 = (b64 or i8) and Lq64 is not None    # both projections accumulate in float64: float64 partials
        T32, T = T, (T64 if p64 else max(T, T64))
        # tile rows one of the two kernels does not fill (their tile heights differ for some shapes) stay zero
        part = (torch.zeros if ((T64 != T32 and not p64) or (i8 and p64)) else torch.empty)(
            (3, batch, max(T, 1), n), dtype=torch.float64 if p64 else ref.dtype, device=ref.device)
        if i8:
            _project_a_i8(W64f, kZ, kx, kls, kos, m, A, part[0], part[1], T, p64, 5 if i8_planes == 5 else 4, flops,
                          kzx_out=i8_kzx_out)
        elif b64 and p64:
            _timed(lambda: _lib.call('nsgp_svgp_tri_gemm_colstats_f64acc_b64', _p(W64f), _p(Kzx64), _p(m), batch, M, n, _p(A),
                                     _p(part[0]), _p(part[1]), T, st), flops, 'f64acc')
        elif b64:                                 # float64 Kzx, float32 partials (the second projection stays float32)
            _timed(lambda: _lib.call('nsgp_svgp_tri_gemm_colstats_f64acc_b64p32', _p(W64f), _p(Kzx64), _p(m), batch, M, n, _p(A),
                                     _p(part[0]), _p(part[1]), T, st), flops, 'f64acc')
        elif fused:
            _timed(lambda: _lib.call('nsgp_svgp_kzx_gemm_colstats_f64acc', _p(W64f), _p(kZ), _p(kx),
                                     n * D if kx.dim() == 3 else 0, _p(kls), _p(kos), D, _p(m), batch, M, n, _p(A),
                                     _p(part[0]), _p(part[1]), T, st), flops, 'f64acc')
        else:
            _timed(lambda: _lib.call('nsgp_svgp_tri_gemm_colstats_f64acc', _p(W64f), _p(Kzx), _p(m), batch, M, n, _p(A),
                                     _p(part[0]), _p(part[1]), T, st), flops, 'f64acc')
    else:
        part = torch.empty((3, batch, max(T, 1), n), dtype=ref.dtype, device=ref.device)
        _timed(lambda: _lib.call(f'nsgp_svgp_tri_gemm_colstats_{sfx}', _p(W), 0, _p(Kzx), _p(m), batch, M, n, _p(A),
                                 _p(part[0]), _p(part[1]), st), flops, ref.dtype)
    if Lq64 is not None:                      # C = Lq^T A accumulated in float64 (layers that feed the next layer)
        if W64f is None or not (b64 or i8) or Lq64.dtype != torch.float64 or Lq64.shape != (batch, M, M) or Lq64.device != ref.device:
            raise BackendError('svgp_project: Lq64 must be the float64 (b,M,M) copy of Lq (with W64f and Kzx64 / i8_inputs)')
        _timed(lambda: _lib.call('nsgp_svgp_tri_gemm_colstats_f64acc_t', _p(_c(Lq64)), _p(A), batch, M, n, _p(C), _p(part[2]),
                                 T, st), flops, 'f64acc')
    else:
        _timed(lambda: _lib.call(f'nsgp_svgp_tri_gemm_colstats_rows_{sfx}', _p(Lq), 1, _p(A), None, batch, M, n, _p(C),
                                 None, _p(part[2]), T, st), flops, ref.dtype)
    mean = torch.empty((batch, n), dtype=ref.dtype, device=ref.device)
    var = torch.empty_like(mean)
    if affine is None:
        x = w = c = None
        sxb = D = swb = scb = 0
    else:
        x, sxb, D, w, swb, c, scb = _affine_args(affine, batch, n, ref)
    fin = 'nsgp_svgp_colstats_finalize_affine_p64_f32' if part.dtype != ref.dtype else f'nsgp_svgp_colstats_finalize_affine_{sfx}'
    _lib.call(fin, _p(part[0]), _p(part[1]), _p(part[2]), _p(base),
              float(base_add), batch, T, n, _p(x), sxb, D, _p(w), swb, _p(c), scb, _p(mean), _p(var), st)
    return A, C, mean, var


def svgp_project_bf16(W, Kzx, Lq, m, base, base_add=0.0, affine=None, W64f=None, kernel_inputs=None, i8_inputs=None,
                      i8_kzx_out=None):
    """BASELINE configs[4]'s "bf16 forward": the forward projections of a float32 SVGP layer with bf16 matrix-core products
    (bf16 operands, float32 accumulation, float32 A / C; csrc/gemm_bf16.hip), column statistics in the epilogues.

    kernel_inputs=None (settings.forward_precision('bf16')): A = W Kzx runs as in `svgp_project` (float32, or float64
        accumulation with W64f) -- its terms |W||Kzx| ~ 1e2 cancel to O(1), which bf16 operands cannot carry (measured at
        M = 2048: 160 % error on the layer outputs) -- and C = Lq^T A, whose operands are O(1), runs on the bf16 cores from a
        bf16 transposed copy of A.
    kernel_inputs=(Z, x, ls, os) (forward_precision('bf16_all')): BOTH products in bf16, Kxz written in bf16 by the build
        kernel (nsgp_rbf_build_t_bf16) -- configs[4] to the letter; a throughput figure, not a usable numerical mode.
    Returns A, C, mean, var (float32).  M must be a multiple of 8."""
    ref = _chk(Lq, m, base, Kzx)
    if ref.dtype != torch.float32:
        raise BackendError('svgp_project_bf16: float32 layers only')
    Lq, m, base = _c(Lq), _c(m), _c(base.reshape(-1))
    if Kzx is None:                          # mode 'bf16' with product 1 on the int8 cores: Kzx is never materialised
        if i8_inputs is None or W64f is None or kernel_inputs is not None:
            raise BackendError('svgp_project_bf16: Kzx=None needs i8_inputs and W64f (mode bf16)')
        batch, M = i8_inputs[0].shape[0], i8_inputs[0].shape[1]
        n = i8_inputs[1].shape[-2]
    else:
        Kzx = _c(Kzx)
        batch, M, n = Kzx.shape
    if M % 8 != 0:
        raise BackendError('svgp_project_bf16: M must be a multiple of 8')
    if Lq.shape != (batch, M, M) or m.shape != (batch, M) or base.shape != (batch,):
        raise BackendError('svgp_project_bf16: shapes')
    lib = _lib.load()
    st = _stream()
    bf = torch.bfloat16
    Ub = torch.empty((batch, M, M), dtype=bf, device=ref.device)
    AT = torch.empty((batch, n, M), dtype=bf, device=ref.device)
    _lib.call('nsgp_cast_sq_bf16_f32', _p(Lq), _p(Ub), M, batch, 1, 1, st)          # tril(Lq)^T
    T = int(lib.nsgp_svgp_bf16_tiles(M))                                              # 128-row tiles
    C = torch.empty((batch, M, n), dtype=ref.dtype, device=ref.device)
    flops = 1.0 * M * M * n * batch
    if kernel_inputs is None:
        # product 1 at full precision.  Its partials are laid out with ITS kernel's tile rows: the float32 plan's (Tf), or
        # the float64-accumulating kernel's (T64: 128-row tiles, 64-row ones when the 128-row grid is under one round --
        # e.g. M = 1024, n = 4032, b = 1 gives Tf = 8 but T64 = 16).  One buffer with the largest count; rows a kernel does
        # not fill stay zero, exactly as in svgp_project.
        Tf = int(lib.nsgp_svgp_colstats_tiles(M, n, batch, 4))
        use_i8 = i8_inputs is not None and W64f is not None
        T64 = (int(lib.nsgp_i8_tiles(M)) if use_i8 else int(lib.nsgp_svgp_f64acc_tiles_for(M, n, batch))) if W64f is not None else Tf
        T1 = T64 if W64f is not None else Tf                       # tile rows product 1 writes
        Tp = max(T1, T)
        A = torch.empty((batch, M, n), dtype=ref.dtype, device=ref.device)
        part = (torch.zeros if (T1 != Tp or T != Tp) else torch.empty)((3, batch, max(Tp, 1), n), dtype=ref.dtype,
                                                                       device=ref.device)
        W = _c(W)
        if use_i8:
            kZ, kx, kls, kos = i8_inputs
            _project_a_i8(_c(W64f), _c(kZ), _c(kx), _c(kls), _c(kos.reshape(-1)), m, A, part[0], part[1], Tp, False, 4, flops,
                          kzx_out=i8_kzx_out)
        elif W64f is not None:
            _timed(lambda: _lib.call('nsgp_svgp_tri_gemm_colstats_f64acc', _p(_c(W64f)), _p(Kzx), _p(m), batch, M, n, _p(A),
                                     _p(part[0]), _p(part[1]), Tp, st), flops, 'f64acc')
        elif Tp == Tf:
            _timed(lambda: _lib.call('nsgp_svgp_tri_gemm_colstats_f32', _p(W), 0, _p(Kzx), _p(m), batch, M, n, _p(A),
                                     _p(part[0]), _p(part[1]), st), flops, ref.dtype)
        else:
            _timed(lambda: _lib.call('nsgp_svgp_tri_gemm_colstats_rows_f32', _p(W), 0, _p(Kzx), _p(m), batch, M, n, _p(A),
                                     _p(part[0]), _p(part[1]), Tp, st), flops, ref.dtype)
        _lib.call('nsgp_transpose_cast_bf16', _p(A), _p(AT), batch, M, n, st)
        if Tp != T:
            # the bf16 kernel lays its partials out with its own tile-row count: give it a compact buffer, then widen
            p2 = torch.empty((batch, T, n), dtype=ref.dtype, device=ref.device)
            _timed(lambda: _lib.call('nsgp_svgp_tri_gemm_colstats_bf16', _p(Ub), 2, _p(AT), None, batch, M, n, _p(C), None,
                                     None, _p(p2), st), flops, 'bf16')
            part[2, :, :T].copy_(p2)
        else:
            _timed(lambda: _lib.call('nsgp_svgp_tri_gemm_colstats_bf16', _p(Ub), 2, _p(AT), None, batch, M, n, _p(C), None,
                                     None, _p(part[2]), st), flops, 'bf16')
        T = Tp
    else:
        Z, x, ls, os_ = kernel_inputs
        Z, ls, os_ = _c(Z), _c(ls), _c(os_.reshape(-1))
        D = Z.shape[-1]
        x = _c(x)
        sxb = 0 if x.dim() == 2 else x.shape[1] * D
        Wsrc = _c(W64f) if W64f is not None else _c(W)
        Wb = torch.empty((batch, M, M), dtype=bf, device=ref.device)
        Kxz = torch.empty((batch, n, M), dtype=bf, device=ref.device)
        _lib.call('nsgp_cast_sq_bf16_f64' if Wsrc.dtype == torch.float64 else 'nsgp_cast_sq_bf16_f32', _p(Wsrc), _p(Wb), M,
                  batch, 0, 1, st)
        _lib.call('nsgp_rbf_build_t_bf16', _p(Z), _p(x), _p(ls), _p(os_), batch, M, n, D, sxb, _p(Kxz), st)
        part = torch.empty((3, batch, max(T, 1), n), dtype=ref.dtype, device=ref.device)
        A = torch.empty_like(Kzx)
        _timed(lambda: _lib.call('nsgp_svgp_tri_gemm_colstats_bf16', _p(Wb), 1, _p(Kxz), _p(m), batch, M, n, _p(A), _p(AT),
                                 _p(part[0]), _p(part[1]), st), flops, 'bf16')
        _timed(lambda: _lib.call('nsgp_svgp_tri_gemm_colstats_bf16', _p(Ub), 2, _p(AT), None, batch, M, n, _p(C), None,
                                 None, _p(part[2]), st), flops, 'bf16')
    mean = torch.empty((batch, n), dtype=ref.dtype, device=ref.device)
    var = torch.empty_like(mean)
    if affine is None:
        xa = w = c = None
        sxa = Da = swb = scb = 0
    else:
        xa, sxa, Da, w, swb, c, scb = _affine_args(affine, batch, n, ref)
    _lib.call('nsgp_svgp_colstats_finalize_affine_f32', _p(part[0]), _p(part[1]), _p(part[2]), _p(base), float(base_add),
              batch, T, n, _p(xa), sxa, Da, _p(w), swb, _p(c), scb, _p(mean), _p(var), st)
    return A, C, mean, var


def svgp_project_bwd(Lq, m, A, C, gmean, gvar, affine=None):
    """Adjoints of svgp_project w.r.t. A (total, through C as well), Lq and m:
    Abar = 2 (Lq C) diag(gvar) + m gmean^T - 2 A diag(gvar);  Lqbar = tril(A diag(2 gvar) C^T);  mbar = A gmean.
    Also returns basebar = rowsum(gvar):(b,) and, with `affine` = (x, w, c) as in svgp_project, the gradients
    (wbar, cbar) of the affine prior mean in the shapes of w / c (None where w / c is None)."""
    ref = _chk(Lq, m, A, C, gmean, gvar)
    Lq, m, A, C, gmean, gvar = _c(Lq), _c(m), _c(A), _c(C), _c(gmean), _c(gvar)
    batch, M, n = A.shape
    if C.shape != A.shape or Lq.shape != (batch, M, M) or m.shape != (batch, M) or gmean.shape != (batch, n) \
            or gvar.shape != (batch, n):
        raise BackendError('svgp_project_bwd: shapes')
    lib = _lib.load()
    sfx, st = _sfx(ref), _stream()
    Abar = torch.empty_like(A)
    sink = grad_sink(Lq)                     # the optimiser's gradient bucket, when Lq is a registered parameter
    Lqbar, lq_beta = (torch.empty_like(Lq), 0.0) if sink is None else (sink[0], 1.0 if sink[1] else 0.0)
    mbar = torch.empty_like(m)
    flops = 1.0 * M * M * n * batch
    basebar = torch.empty(batch, dtype=ref.dtype, device=ref.device)
    wbar = cbar = None
    if affine is None:
        x, sxb, D, shared = None, 0, 0, 0
    else:
        x, sxb, D, w, swb, c, scb = _affine_args(affine, batch, n, ref)
        shared = int(swb == 0 and scb == 0)
        if w is not None and c is not None and batch > 1 and (swb == 0) != (scb == 0):
            raise BackendError('affine prior mean: weights and constant must both be shared or both be per batch')
        nb = 1 if shared else batch
        if w is not None:
            wbar = torch.empty((nb, D), dtype=ref.dtype, device=ref.device)
        if c is not None:
            cbar = torch.empty(nb, dtype=ref.dtype, device=ref.device)
    _lib.call(f'nsgp_rowdot_affine_{sfx}', _p(A), _p(gmean), _p(gvar), _p(x) if wbar is not None else None, sxb, D,
              shared, batch, M, n, _p(mbar), _p(basebar), _p(wbar), _p(cbar), st)
    _timed(lambda: _lib.call(f'nsgp_svgp_abar_{sfx}', _p(Lq), _p(C), _p(A), _p(m), _p(gmean), _p(gvar), batch, M, n,
                             _p(Abar), st), flops, ref.dtype)
    wsb = lib.nsgp_svgp_lqbar_workspace(batch, M, n, ref.element_size())
    ws = _ws(wsb, ref.device) if wsb else None
    _timed(lambda: _lib.call(f'nsgp_svgp_lqbar_acc_{sfx}', _p(A), _p(C), _p(gvar), batch, M, n, lq_beta, _p(Lqbar), _p(ws),
                             ws.numel() if ws is not None else 0, st), flops, ref.dtype)
    # accumulated onto a gradient that is already p.grad: nothing to hand to autograd for Lq
    return Abar, (None if lq_beta else Lqbar), mbar, basebar, wbar, cbar


def dgp_sample(mean, var, eps):
    """h[s,i,c] = mean[c,s',i] + sqrt(var[c,s',i]) eps[s,i,c]; mean/var:(b,ns,n) ns in {1,S}; eps:(S,n,b)."""
    ref = _chk(mean, var, eps)
    mean, var, eps = _c(mean), _c(var), _c(eps)
    S, n, b = eps.shape
    ns = mean.shape[1]
    if mean.shape != (b, ns, n) or var.shape != mean.shape or ns not in (1, S):
        raise BackendError('dgp_sample: shapes')
    h = torch.empty_like(eps)
    _lib.call(f'nsgp_dgp_sample_fwd_{_sfx(ref)}', _p(mean), _p(var), _p(eps), S, ns, n, b, _p(h), _stream())
    return h


def dgp_sample_bwd(var, eps, gh):
    ref = _chk(var, eps, gh)
    var, eps, gh = _c(var), _c(eps), _c(gh)
    S, n, b = eps.shape
    ns = var.shape[1]
    gmean, gvar = torch.empty_like(var), torch.empty_like(var)
    _lib.call(f'nsgp_dgp_sample_bwd_{_sfx(ref)}', _p(var), _p(eps), _p(gh), S, ns, n, b, _p(gmean), _p(gvar),
              _stream())
    return gmean, gvar


def _red_ws(ref):
    return _ws(_lib.load().nsgp_reduce_workspace(0, ref.element_size()), ref.device)


def gauss_ell(y, mu, v, noise, scale):
    """out[s] = scale * sum_i E_q log N(y_i | f_si, noise);  mu, v:(S,n)  y:(n,)  noise: 1-element tensor."""
    ref = _chk(y, mu, v, noise)
    y, mu, v = _c(y), _c(mu), _c(v)
    S, n = mu.shape
    if y.shape != (n,) or v.shape != mu.shape:
        raise BackendError('gauss_ell: shapes')
    out = torch.empty(S, dtype=ref.dtype, device=ref.device)
    ws = _red_ws(ref)
    _lib.call(f'nsgp_gauss_ell_fwd_{_sfx(ref)}', _p(y), _p(mu), _p(v), _p(noise.reshape(1)), S, n, float(scale),
              _p(out), _p(ws), ws.numel(), _stream())
    return out


def gauss_ell_bwd(y, mu, v, noise, scale, gout, need_noise=True):
    ref = _chk(y, mu, v, noise, gout)
    y, mu, v, gout = _c(y), _c(mu), _c(v), _c(gout)
    S, n = mu.shape
    if gout.shape != (S,):
        raise BackendError('gauss_ell_bwd: gout must be (S,)')
    gmu, gv = torch.empty_like(mu), torch.empty_like(mu)
    gn = torch.empty(1, dtype=ref.dtype, device=ref.device) if need_noise else None
    ws = _red_ws(ref)
    _lib.call(f'nsgp_gauss_ell_bwd_{_sfx(ref)}', _p(y), _p(mu), _p(v), _p(noise.reshape(1)), S, n, float(scale),
              _p(gout), _p(gmu), _p(gv), _p(gn), _p(ws), ws.numel(), _stream())
    return gmu, gv, gn


def kl_whitened(m, Lq):
    ref = _chk(m, Lq)
    m, Lq = _c(m), _c(Lq)
    if m.dim() == 1:
        m, Lq = m.unsqueeze(0), Lq.unsqueeze(0)
    batch, M = m.shape
    if Lq.shape != (batch, M, M):
        raise BackendError('kl_whitened: shapes')
    out = torch.empty(batch, dtype=ref.dtype, device=ref.device)
    ws = _red_ws(ref)
    _lib.call(f'nsgp_kl_whitened_fwd_{_sfx(ref)}', _p(m), _p(Lq), batch, M, _p(out), _p(ws), ws.numel(), _stream())
    return out


def kl_whitened_bwd(m, Lq, gout):
    ref = _chk(m, Lq)
    shp_m, shp_L = m.shape, Lq.shape
    m, Lq = _c(m), _c(Lq)
    if m.dim() == 1:
        m, Lq = m.unsqueeze(0), Lq.unsqueeze(0)
    batch, M = m.shape
    gm, gL = torch.empty_like(m), torch.empty_like(Lq)
    _lib.call(f'nsgp_kl_whitened_bwd_{_sfx(ref)}', _p(m), _p(Lq), batch, M, float(gout), _p(gm), _p(gL), _stream())
    return gm.reshape(shp_m), gL.reshape(shp_L)


def philox_normal(seed, stream_id, row0, S, n, b, dtype=torch.float32, device='cuda', step_dev=None):
    """eps:(S,n,b) standard normals keyed by (seed, stream_id, global row, sample, column).  `step_dev`
    (1-element int64 device tensor) replaces the high word of stream_id on the device (graph replays)."""
    eps = torch.empty((S, n, b), dtype=dtype, device=device)
    if not eps.is_cuda:
        raise BackendError('philox_normal: CUDA device required')
    with torch.cuda.device(eps.device):
        _lib.call(f'nsgp_philox_normal_{_sfx(eps)}', ctypes.c_uint64(seed), ctypes.c_uint64(stream_id),
                  _p(step_dev), row0, S, n, b, _p(eps), _stream())
    return eps


def adam_step_(p, g, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, step, grad_scale=1.0, step_dev=None):
    _chk(p, g, exp_avg, exp_avg_sq)
    if p.dtype != torch.float32 or not all(t.is_contiguous() for t in (p, g, exp_avg, exp_avg_sq)):
        raise BackendError('adam_step_: contiguous float32 flat buffers expected')
    _lib.call('nsgp_adam_step_f32', _p(p), _p(g), _p(exp_avg), _p(exp_avg_sq), p.numel(), float(lr), float(beta1),
              float(beta2), float(eps), int(step), _p(step_dev), float(grad_scale), _stream())
    return p


# --------------------------------------------------------------------------------------------
# autograd Functions
# --------------------------------------------------------------------------------------------
class GibbsKernelFn(torch.autograd.Function):
    """K = os * Gibbs(x1,x2;ell1,ell2) + diag_add*I  (GibbsKernel.forward under GibbsSafeScaleKernel,
    models/gibbs_kernels.py:135-168).  Pass the same tensor as ell1 and ell2 for K_xx: autograd then
    sums both roles."""

    @staticmethod
    def forward(ctx, x1, x2, ell1, ell2, outputscale, diag_add):
        ctx.save_for_backward(x1, x2, ell1, ell2, outputscale)
        ctx.has_diag = diag_add is not None
        ctx.diag_shape = diag_add.shape if torch.is_tensor(diag_add) else None
        return gibbs_build(x1, x2, ell1, ell2, outputscale, diag_add)

    @staticmethod
    def backward(ctx, G):
        x1, x2, ell1, ell2, outputscale = ctx.saved_tensors
        need_x = ctx.needs_input_grad[0] or ctx.needs_input_grad[1]
        g_l1, g_l2, g_x1, g_x2, g_os = gibbs_build_bwd(x1, x2, ell1, ell2, outputscale, G, need_x=need_x)
        g_diag = None
        if ctx.has_diag and ctx.needs_input_grad[5]:
            g_diag = torch.diagonal(G).sum().reshape(ctx.diag_shape if ctx.diag_shape is not None else ())
        g_osr = None
        if outputscale is not None and ctx.needs_input_grad[4]:
            g_osr = g_os.reshape(outputscale.shape)
        return (g_x1 if ctx.needs_input_grad[0] else None, g_x2 if ctx.needs_input_grad[1] else None,
                g_l1 if ctx.needs_input_grad[2] else None, g_l2 if ctx.needs_input_grad[3] else None,
                g_osr, g_diag)


class RbfKernelFn(torch.autograd.Function):
    """K[b] = os[b] RBF-ARD(x1, x2; ls[b]) + diag_add I   (gpytorch ScaleKernel(RBFKernel))."""

    @staticmethod
    def forward(ctx, x1, x2, ls, os_, diag_add):
        ctx.save_for_backward(x1, x2, ls, os_)
        return rbf_build(x1, x2, ls, os_, diag_add)

    @staticmethod
    def backward(ctx, G):
        x1, x2, ls, os_ = ctx.saved_tensors
        n1g, n2g = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        g_x1, g_x2, g_ls, g_os = rbf_build_bwd(x1, x2, ls, os_, G, need_x1=n1g, need_x2=n2g)
        if n1g and x1.dim() == 2:
            g_x1 = g_x1.sum(0)
        if n2g and x2.dim() == 2:
            g_x2 = g_x2.sum(0)
        return (g_x1 if n1g else None, g_x2 if n2g else None, g_ls.reshape(ls.shape), g_os.reshape(os_.shape), None)


class RbfPeriodicKernelFn(torch.autograd.Function):
    """K[b] = os[b] RBF-ARD(x; ls_rbf[b]) Periodic(x; ls_per[b], period[b]) + diag_add I; ls_rbf / os may be None
    (gpytorch ScaleKernel(RBFKernel * PeriodicKernel), models/spatio_temporal_models.py:22,42)."""

    @staticmethod
    def forward(ctx, x1, x2, ls_rbf, ls_per, period, os_, diag_add):
        ctx.save_for_backward(x1, x2, ls_per, period, *([ls_rbf] if ls_rbf is not None else []),
                              *([os_] if os_ is not None else []))
        ctx.has = (ls_rbf is not None, os_ is not None)
        return rbf_periodic_build(x1, x2, ls_rbf, ls_per, period, os_, diag_add)

    @staticmethod
    def backward(ctx, G):
        x1, x2, ls_per, period, *rest = ctx.saved_tensors
        ls_rbf = rest.pop(0) if ctx.has[0] else None
        os_ = rest.pop(0) if ctx.has[1] else None
        n1g, n2g = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        g_x1, g_x2, g_lr, g_lp, g_pe, g_os = rbf_periodic_build_bwd(x1, x2, ls_rbf, ls_per, period, os_, G, n1g, n2g)
        if n1g and x1.dim() == 2:
            g_x1 = g_x1.sum(0)
        if n2g and x2.dim() == 2:
            g_x2 = g_x2.sum(0)
        return (g_x1 if n1g else None, g_x2 if n2g else None,
                g_lr.reshape(ls_rbf.shape) if ls_rbf is not None else None, g_lp.reshape(ls_per.shape),
                g_pe.reshape(period.shape), g_os.reshape(os_.shape) if os_ is not None else None, None)


class Ps2dKernelFn(torch.autograd.Function):
    """Paciorek-Schervish kernel for per-point 2x2 matrices (models/multivariate_gibbs_kernel.py:98-150)."""

    @staticmethod
    def forward(ctx, x1, x2, s1, s2, jitter):
        ctx.save_for_backward(x1, x2, s1, s2)
        ctx.jitter = jitter
        return ps2d_build(x1, x2, s1, s2, jitter)

    @staticmethod
    def backward(ctx, G):
        x1, x2, s1, s2 = ctx.saved_tensors
        g1, g2 = ps2d_build_bwd(x1, x2, s1, s2, ctx.jitter, G)
        return None, None, g1 if ctx.needs_input_grad[2] else None, g2 if ctx.needs_input_grad[3] else None, None


def _tri_flags(ta, tb, a_lower, b_lower, out_lower):
    f = 0
    if a_lower:
        f |= GEMM_A_UPPER if ta else GEMM_A_LOWER
    if b_lower:
        f |= GEMM_B_UPPER if tb else GEMM_B_LOWER
    if out_lower:
        f |= GEMM_C_LOWER
    return f


class MatmulFn(torch.autograd.Function):
    """C = op(A) op(B) with optional lower-triangular structure of the *stored* A / B / C.
    Differentiable to any order (backward recurses through `matmul`)."""

    @staticmethod
    def forward(ctx, A, B, ta, tb, a_lower, b_lower, out_lower):
        ctx.save_for_backward(A, B)
        ctx.cfg = (ta, tb, a_lower, b_lower, out_lower)
        return gemm(A, B, ta, tb, flags=_tri_flags(ta, tb, a_lower, b_lower, out_lower))

    @staticmethod
    def backward(ctx, G):
        A, B = ctx.saved_tensors
        ta, tb, a_lower, b_lower, out_lower = ctx.cfg
        gA = gB = None
        if ctx.needs_input_grad[0]:
            if not ta:
                gA = matmul(G, B, False, not tb, out_lower, b_lower, a_lower)
            else:
                gA = matmul(B, G, tb, True, b_lower, out_lower, a_lower)
            if A.dim() == 2 and gA.dim() == 3:
                gA = gA.sum(0)
        if ctx.needs_input_grad[1]:
            if not tb:
                gB = matmul(A, G, not ta, False, a_lower, out_lower, b_lower)
            else:
                gB = matmul(G, A, True, ta, out_lower, a_lower, b_lower)
            if B.dim() == 2 and gB.dim() == 3:
                gB = gB.sum(0)
        return gA, gB, None, None, None, None, None


def matmul(A, B, ta=False, tb=False, a_lower=False, b_lower=False, out_lower=False):
    return MatmulFn.apply(A, B, ta, tb, a_lower, b_lower, out_lower)


class CholInvFn(torch.autograd.Function):
    """W = chol(K)^-1 (lower), so that K^-1 = W^T W and log|K| = -2 sum log diag W.

    Replaces psd_safe_cholesky + triangular_solve(eye, .) (models/gibbs_kernels.py:197-208) and the
    Cholesky factor / inv_matmul of gpytorch's VariationalStrategy and ExactMarginalLogLikelihood.
    Backward:  Kbar = -W^T sym(Phi(tril(Wbar) W^T)) W  with sym(Phi(B)) = (Phi(B) + Phi(B)^T)/2.
    """

    @staticmethod
    def forward(ctx, K):
        L, info = potrf(K)
        W = trtri(L)
        ctx.save_for_backward(W)
        ctx.mark_non_differentiable(info)
        return W, info

    @staticmethod
    def backward(ctx, Wbar, _):
        (W,) = ctx.saved_tensors
        Bm = gemm(Wbar, W, False, True, flags=GEMM_A_LOWER | GEMM_B_UPPER)      # tril(Wbar) W^T
        S = chol_bwd_phi_sym(Bm)                                                 # Phi + Phi^T
        T = gemm(S, W, False, False, flags=GEMM_B_LOWER)
        return gemm(W, T, True, False, alpha=-0.5, flags=GEMM_A_UPPER)


def chol_inv(K):
    """Returns (W, info): W lower with K^-1 = W^T W; info int32 per matrix (0 = ok)."""
    return CholInvFn.apply(K)


def gibbs_kernel(x1, x2, ell1, ell2, outputscale=None, diag_add=None):
    return GibbsKernelFn.apply(x1, x2, ell1, ell2, outputscale, diag_add)


def rbf_kernel(x1, x2, ls, os_, diag_add=0.0):
    """Batched: returns (batch, n1, n2); ls:(batch,D) os:(batch,)."""
    return RbfKernelFn.apply(x1, x2, ls, os_, diag_add)


def rbf_periodic_kernel(x1, x2, ls_rbf, ls_per, period, os_=None, diag_add=0.0):
    """Batched (batch, n1, n2): os * RBF-ARD(ls_rbf) * Periodic(ls_per, period); ls_rbf / os_ None = factor absent."""
    return RbfPeriodicKernelFn.apply(x1, x2, ls_rbf, ls_per, period, os_, diag_add)


def ps2d_kernel(x1, x2, s1, s2, jitter=1e-5):
    return Ps2dKernelFn.apply(x1, x2, s1, s2, jitter)


class GaussEllFn(torch.autograd.Function):
    """(S,) vector: scale * sum_i E_q log N(y_i | f_si, noise)   (GaussianLikelihood.expected_log_prob
    summed over the minibatch, per likelihood sample)."""

    @staticmethod
    def forward(ctx, y, mu, v, noise, scale):
        ctx.save_for_backward(y, mu, v, noise)
        ctx.scale = scale
        return gauss_ell(y, mu, v, noise, scale)

    @staticmethod
    def backward(ctx, g):
        y, mu, v, noise = ctx.saved_tensors
        gmu, gv, gn = gauss_ell_bwd(y, mu, v, noise, ctx.scale, g, need_noise=ctx.needs_input_grad[3])
        return None, gmu, gv, gn.reshape(noise.shape) if gn is not None else None, None


class KlWhitenedFn(torch.autograd.Function):
    """sum_b KL(N(m_b, Lq_b Lq_b^T) || N(0, I))   (whitened VariationalStrategy.kl_divergence)."""

    @staticmethod
    def forward(ctx, m, Lq):
        ctx.save_for_backward(m, Lq)
        return kl_whitened(m, Lq).sum()

    @staticmethod
    def backward(ctx, g):
        m, Lq = ctx.saved_tensors
        gm, gL = kl_whitened_bwd(m, Lq, 1.0)
        return gm * g, gL * g


class GaussEllTotalFn(torch.autograd.Function):
    """Scalar  scale * sum_s sum_i E_q log N(y_i | f_si, noise): the (S,) vector of GaussEllFn followed by a
    mean / scaling, as ONE reduction; its backward reads the upstream gradient on the device."""

    @staticmethod
    def forward(ctx, y, mu, v, noise, scale):
        ref = _chk(y, mu, v, noise)
        y, mu, v = _c(y), _c(mu), _c(v)
        S, n = mu.shape
        if y.shape != (n,) or v.shape != mu.shape:
            raise BackendError('gauss_ell_total: shapes')
        out = torch.empty(1, dtype=ref.dtype, device=ref.device)
        ws = _red_ws(ref)
        _lib.call(f'nsgp_gauss_ell_total_fwd_{_sfx(ref)}', _p(y), _p(mu), _p(v), _p(noise.reshape(1)), S, n, float(scale),
                  _p(out), _p(ws), ws.numel(), _stream())
        ctx.save_for_backward(y, mu, v, noise)
        ctx.scale = float(scale)
        return out.reshape(())

    @staticmethod
    def backward(ctx, g):
        y, mu, v, noise = ctx.saved_tensors
        S, n = mu.shape
        gmu, gv = torch.empty_like(mu), torch.empty_like(mu)
        need_noise = ctx.needs_input_grad[3]
        gn = torch.empty(1, dtype=mu.dtype, device=mu.device) if need_noise else None
        ws = _red_ws(mu)
        _lib.call(f'nsgp_gauss_ell_total_bwd_{_sfx(mu)}', _p(y), _p(mu), _p(v), _p(noise.reshape(1)), S, n, ctx.scale,
                  _p(_c(g).reshape(1)), _p(gmu), _p(gv), _p(gn), _p(ws), ws.numel(), _stream())
        return None, gmu, gv, gn.reshape(noise.shape) if gn is not None else None, None


class KlWhitenedTotalFn(torch.autograd.Function):
    """Scalar  addin + scale * sum_b KL(N(m_b, Lq_b Lq_b^T) || N(0, I))  (addin: optional scalar tensor, so the terms of
    an objective chain without separate additions); backward reads the upstream gradient on the device."""

    @staticmethod
    def forward(ctx, m, Lq, scale, addin=None):
        ref = _chk(m, Lq)
        m2, L2 = _c(m), _c(Lq)
        if m2.dim() == 1:
            m2, L2 = m2.unsqueeze(0), L2.unsqueeze(0)
        batch, M = m2.shape
        if L2.shape != (batch, M, M) or batch == 0:
            raise BackendError('kl_whitened_total: shapes')
        if addin is not None:
            _chk(ref, addin)
            if addin.numel() != 1:
                raise BackendError('kl_whitened_total: addin must be a scalar')
        out = torch.empty(1, dtype=ref.dtype, device=ref.device)
        ws = _red_ws(ref)
        _lib.call(f'nsgp_kl_whitened_total_acc_fwd_{_sfx(ref)}', _p(m2), _p(L2), batch, M, float(scale),
                  None if addin is None else _p(_c(addin).reshape(1)), _p(out), _p(ws), ws.numel(), _stream())
        ctx.save_for_backward(m2, L2)
        ctx.scale, ctx.shapes, ctx.has_addin = float(scale), (m.shape, Lq.shape), addin is not None
        return out.reshape(())

    @staticmethod
    def backward(ctx, g):
        m2, L2 = ctx.saved_tensors
        batch, M = m2.shape
        sink = grad_sink(L2)                 # first writer of the gradient bucket's range for Lq (see grad_sink)
        gm = torch.empty_like(m2)
        gL = sink[0] if (sink is not None and not sink[1]) else torch.empty_like(L2)
        _lib.call(f'nsgp_kl_whitened_total_bwd_{_sfx(m2)}', _p(m2), _p(L2), batch, M, ctx.scale, _p(_c(g).reshape(1)),
                  _p(gm), _p(gL), _stream())
        return gm.reshape(ctx.shapes[0]), gL.reshape(ctx.shapes[1]), None, (g if ctx.has_addin else None)


class DsviObjectiveFn(torch.autograd.Function):
    """Scalar  ell_scale * sum_s sum_i E_q log N(y_i | f_si, noise) + kl_scale * sum_g sum_b KL(N(m_gb, Lq_gb Lq_gb^T) || N(0, I))
    for a list of variational groups (m_g:(b_g,M) or (M,), Lq_g:(b_g,M,M) or (M,M); one per layer, all of the same M):
    nsgp_dsvi_objective_{fwd,bwd} -- two launches forward, two backward, whatever the number of layers (the chain of
    GaussEllTotalFn + one KlWhitenedTotalFn per layer it replaces: 2 + 2 per layer forward, 3 + 1 per layer backward).
    apply(y, mu, v, noise, ell_scale, kl_scale, m_0, Lq_0, m_1, Lq_1, ...)"""

    @staticmethod
    def _arrays(ms, Ls):
        ng = len(ms)
        pm = (ctypes.c_void_p * ng)(*[t.data_ptr() for t in ms])
        pL = (ctypes.c_void_p * ng)(*[t.data_ptr() for t in Ls])
        nb = (ctypes.c_int64 * ng)(*[t.shape[0] for t in ms])
        return ng, pm, pL, nb

    @staticmethod
    def forward(ctx, y, mu, v, noise, ell_scale, kl_scale, *mL):
        ref = _chk(y, mu, v, noise, *mL)
        y, mu, v = _c(y), _c(mu), _c(v)
        S, n = mu.shape
        if y.shape != (n,) or v.shape != mu.shape or len(mL) % 2 or len(mL) > 16:
            raise BackendError('dsvi_objective: shapes')
        ms, Ls = [], []
        for m, Lq in zip(mL[0::2], mL[1::2]):
            m2, L2 = _c(m), _c(Lq)
            if m2.dim() == 1:
                m2, L2 = m2.unsqueeze(0), L2.unsqueeze(0)
            ms.append(m2)
            Ls.append(L2)
        M = ms[0].shape[1] if ms else 0
        if any(m2.shape[1] != M or L2.shape != (m2.shape[0], M, M) for m2, L2 in zip(ms, Ls)):
            raise BackendError('dsvi_objective: every group must be (b,M) / (b,M,M) with one M')
        out = torch.empty(1, dtype=ref.dtype, device=ref.device)
        lib = _lib.load()
        ws = _ws(lib.nsgp_dsvi_objective_workspace(S, n, M, sum(m2.shape[0] for m2 in ms), ref.element_size()), ref.device)
        ng, pm, pL, nb = DsviObjectiveFn._arrays(ms, Ls)
        _lib.call(f'nsgp_dsvi_objective_fwd_{_sfx(ref)}', _p(y), _p(mu), _p(v), _p(noise.reshape(1)), S, n, float(ell_scale), ng,
                  pm, pL, nb, M, float(kl_scale), _p(out), _p(ws), ws.numel(), _stream())
        ctx.save_for_backward(y, mu, v, noise, *ms, *Ls)
        ctx.cfg = (float(ell_scale), float(kl_scale), len(ms), [(m.shape, Lq.shape) for m, Lq in zip(mL[0::2], mL[1::2])])
        return out.reshape(())

    @staticmethod
    def backward(ctx, g):
        ell_scale, kl_scale, ng, shapes = ctx.cfg
        y, mu, v, noise, *rest = ctx.saved_tensors
        ms, Ls = rest[:ng], rest[ng:]
        S, n = mu.shape
        M = ms[0].shape[1] if ms else 0
        gmu, gv = torch.empty_like(mu), torch.empty_like(mu)
        need_noise = ctx.needs_input_grad[3]
        gn = torch.empty(1, dtype=mu.dtype, device=mu.device) if need_noise else None
        gms = [torch.empty_like(m2) for m2 in ms]
        gLs = []
        for L2 in Ls:                          # first writer of the gradient bucket's range for Lq (see grad_sink)
            sink = grad_sink(L2)
            gLs.append(sink[0] if (sink is not None and not sink[1]) else torch.empty_like(L2))
        lib = _lib.load()
        ws = _ws(lib.nsgp_dsvi_objective_workspace(S, n, 0, 0, mu.element_size()), mu.device)
        _, pm, pL, nb = DsviObjectiveFn._arrays(ms, Ls)
        pgm = (ctypes.c_void_p * ng)(*[t.data_ptr() for t in gms])
        pgL = (ctypes.c_void_p * ng)(*[t.data_ptr() for t in gLs])
        _lib.call(f'nsgp_dsvi_objective_bwd_{_sfx(mu)}', _p(y), _p(mu), _p(v), _p(noise.reshape(1)), S, n, ell_scale, ng, pm, pL,
                  nb, M, kl_scale, _p(_c(g).reshape(1)), _p(gmu), _p(gv), _p(gn), pgm, pgL, _p(ws), ws.numel(), _stream())
        grads = []
        for gm, gL, (sm, sL) in zip(gms, gLs, shapes):
            grads += [gm.reshape(sm), gL.reshape(sL)]
        return (None, gmu, gv, gn.reshape(noise.shape) if gn is not None else None, None, None, *grads)


class DgpSampleFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mean, var, eps):
        ctx.save_for_backward(var, eps)
        return dgp_sample(mean, var, eps)

    @staticmethod
    def backward(ctx, gh):
        var, eps = ctx.saved_tensors
        gm, gv = dgp_sample_bwd(var, eps, gh)
        return gm, gv, None
