"""Flat parameter / gradient buckets and a fused Adam step.

The reference optimises with torch.optim.Adam(model.parameters(), lr=0.01)
(experiments/deepgp_spatial_bench.py:74-76).  Here all trainable parameters live in ONE contiguous
float32 buffer (parameters and their .grad are views into it), so that
  * the Adam update is one HIP kernel launch (nsgp_adam_step_f32), and
  * the data-parallel gradient exchange is one RCCL all-reduce of one bucket (nsgp.dist).
torch.optim.Adam keeps working on the same model (the views are ordinary Parameters)."""
import torch

from . import ops


class FlatBucket:
    """Re-homes `params` into one flat buffer; p.data and p.grad become views (device-agnostic).

    `grads_as_views=True` (default): autograd accumulates straight into the bucket (`p.grad += g`, one small
    launch per parameter).  `grads_as_views=False`: `zero_grad()` drops the .grad tensors so autograd hands over
    its own buffers (no launch, no memset), and `gather_grads()` packs them into the bucket with one
    multi-tensor copy -- fewer launches per step; FusedAdam and DataParallel call it themselves.

    Aliasing with `grads_as_views=False` on the GPU: parameters of >= 65,536 elements are registered as gradient sinks
    (`ops.register_grad_sink`), i.e. their backward kernels write the gradient straight into this bucket and autograd
    receives a VIEW of `flat_g`.  After `loss.backward()` that view is `p.grad` (intended).  A gradient obtained with
    `torch.autograd.grad(...)`, or a `p.grad` kept across `zero_grad()`, aliases the bucket too and is overwritten by the
    next backward pass -- take such gradients inside `with nsgp.ops.grad_sinks(False):` (fresh buffers, as plain torch), or
    clone them."""

    def __init__(self, params, grads_as_views=True, stage_of=None):
        seen, plist = set(), []
        for p in params:
            if p.requires_grad and id(p) not in seen:
                seen.add(id(p))
                plist.append(p)
        if not plist:
            raise ValueError('FlatBucket: no trainable parameters')
        # `stage_of` ({id(p): k}, nsgp.stages.BackwardStages.final_stage_of): lay the parameters out by the backward
        # stage that completes their gradient, so every stage's gradients are ONE contiguous range of the bucket
        # (`segments`) and can be exchanged as soon as that stage has run.
        if stage_of is not None:
            plist.sort(key=lambda p: stage_of.get(id(p), 0))          # stable: model order within a stage
        dev, dt = plist[0].device, plist[0].dtype
        for p in plist:
            if p.device != dev or p.dtype != dt:
                raise ValueError('FlatBucket: all parameters must share device and dtype')
        self.params = plist
        self.grads_as_views = grads_as_views
        # every parameter starts on a 256-byte boundary of the bucket: the GEMM / kernel-build loaders take their 16-byte
        # path only for aligned operands (an odd-sized neighbour, e.g. a 1-element mean constant in front of a 1024 x 1024
        # Cholesky factor, would otherwise push every later parameter onto the element-wise edge path: 2.5x slower
        # projections).  The padding elements stay zero (zero gradient -> Adam leaves them at zero).
        ALIGN = 256 // torch.empty((), dtype=dt).element_size()
        starts, off = [], 0
        for p in plist:
            starts.append(off)
            off += (p.numel() + ALIGN - 1) // ALIGN * ALIGN
        self.numel = off                                            # bucket length, padding included
        self.num_param_elements = sum(p.numel() for p in plist)
        self.flat_p = torch.zeros(self.numel, dtype=dt, device=dev)
        self.flat_g = torch.zeros(self.numel, dtype=dt, device=dev)
        self.offsets = []
        with torch.no_grad():
            for p, off in zip(plist, starts):
                n = p.numel()
                self.flat_p[off:off + n].copy_(p.reshape(-1))
                p.data = self.flat_p[off:off + n].view(p.shape)
                p.grad = self.flat_g[off:off + n].view(p.shape) if grads_as_views else None
                self.offsets.append((off, n))
        self.grad_views = [self.flat_g[off:off + n].view(p.shape) for p, (off, n) in zip(plist, self.offsets)]
        # large parameters on the GPU: their backward kernels write the gradient straight into the bucket (ops.grad_sink)
        # instead of a fresh buffer that gather_grads() would copy (the three 1024 x 1024 Cholesky factors of the headline
        # model are 12 of the bucket's 12.6 MB)
        if not grads_as_views and dev.type == 'cuda':
            for p, (off, n) in zip(plist, self.offsets):
                if n >= (1 << 16):
                    ops.register_grad_sink(p, self.flat_g, off)
        self.stage_index = [0 if stage_of is None else int(stage_of.get(id(p), 0)) for p in plist]
        self.segments = {}                                             # stage -> (first element, one past the last)
        for k, (off, n) in zip(self.stage_index, self.offsets):
            end = (off + n + ALIGN - 1) // ALIGN * ALIGN                # the padding travels with its parameter
            a, b = self.segments.get(k, (off, off))
            self.segments[k] = (min(a, off), max(b, end))

    def __del__(self):
        try:
            ops.unregister_grad_sinks(self.flat_g)
        except Exception:
            pass

    def unpadded(self, flat):
        """The parameters' elements of a bucket-shaped buffer (flat_p, flat_g, an Adam moment) without the alignment
        padding, concatenated in bucket order."""
        return torch.cat([flat[off:off + n] for off, n in self.offsets])

    def zero_grad(self):
        if not self.grads_as_views:
            for p in self.params:
                p.grad = None
            return
        self.flat_g.zero_()
        for p, (off, n) in zip(self.params, self.offsets):       # re-attach if something replaced .grad
            if p.grad is None or p.grad.data_ptr() != self.flat_g.data_ptr() + off * self.flat_g.element_size():
                p.grad = self.flat_g[off:off + n].view(p.shape)

    def check_homed(self):
        """Raise if a parameter no longer lives in the flat buffer (model.to() / .double() / .cuda() after the bucket
        was built re-allocates p.data: Adam would then update an orphan buffer).  Pointer compares only, no sync."""
        base, es = self.flat_p.data_ptr(), self.flat_p.element_size()
        for p, (off, n) in zip(self.params, self.offsets):
            if p.data_ptr() != base + off * es or p.dtype != self.flat_p.dtype:
                raise RuntimeError('FlatBucket: a parameter was moved out of the flat buffer (model.to()/.double()/'
                                   '.cuda() after the optimiser was built?); rebuild the FlatBucket / FusedAdam')

    def gather_grads(self, stage=None):
        """Pack the parameters' .grad tensors into the flat gradient buffer (no-op for view gradients); `stage`: only
        the parameters of that backward stage (their range of the bucket is `segments[stage]`)."""
        self.check_homed()
        if self.grads_as_views:
            return
        dst, src, missing = [], [], []
        for p, v, k in zip(self.params, self.grad_views, self.stage_index):
            if stage is not None and k != stage:
                continue
            if p.grad is None:
                missing.append(v)
            elif p.grad.data_ptr() != v.data_ptr():
                dst.append(v)
                src.append(p.grad)
        if dst:
            torch._foreach_copy_(dst, src)
        if missing:
            torch._foreach_zero_(missing)


class FusedAdam:
    """torch.optim.Adam semantics (no weight decay, no amsgrad) over a FlatBucket, one kernel per step.
    `capturable=True` keeps the step count on the device so the update can live in a hipGraph."""

    def __init__(self, params, lr=0.01, betas=(0.9, 0.999), eps=1e-8, capturable=False, grads_as_views=True,
                 stage_of=None):
        self.bucket = params if isinstance(params, FlatBucket) else FlatBucket(list(params), grads_as_views, stage_of)
        if self.bucket.flat_p.dtype != torch.float32:
            raise ValueError('FusedAdam: float32 parameters expected')
        self.lr, self.betas, self.eps = lr, betas, eps
        self.exp_avg = torch.zeros_like(self.bucket.flat_p)
        self.exp_avg_sq = torch.zeros_like(self.bucket.flat_p)
        self.steps = 0
        self.step_dev = torch.zeros(1, dtype=torch.int64, device=self.bucket.flat_p.device) if capturable else None

    @property
    def param_groups(self):
        """torch.optim-style view (read-only use: schedulers / logging read `lr`)."""
        return [{'params': self.bucket.params, 'lr': self.lr, 'betas': self.betas, 'eps': self.eps}]

    def state_dict(self):
        """Plain tensors / numbers only, so harness checkpoints load with torch.load(weights_only=True)
        (the reference's best.tar / final.tar carry 'optim_state', experiments/precipitation_baselines.py:376-397)."""
        return {'step': int(self.steps), 'exp_avg': self.exp_avg.detach().clone(),
                'exp_avg_sq': self.exp_avg_sq.detach().clone(), 'lr': float(self.lr),
                'betas': [float(self.betas[0]), float(self.betas[1])], 'eps': float(self.eps),
                'numel': int(self.bucket.num_param_elements),
                'layout': [[int(off), int(n)] for off, n in self.bucket.offsets]}

    def load_state_dict(self, state):
        """In-place restore: the moment buffers and the device step counter keep their addresses, so a hipGraph
        captured around step() stays valid."""
        if int(state['numel']) != self.bucket.num_param_elements:
            raise ValueError(f"FusedAdam.load_state_dict: {state['numel']} parameters in the checkpoint, "
                             f'{self.bucket.num_param_elements} in the bucket')
        if [list(map(int, x)) for x in state.get('layout', [])] != [[int(off), int(n)] for off, n in self.bucket.offsets]:
            raise ValueError('FusedAdam.load_state_dict: the checkpoint was written with a different bucket layout '
                             '(parameter order / staged-backward grouping)')
        with torch.no_grad():
            self.exp_avg.copy_(state['exp_avg'])
            self.exp_avg_sq.copy_(state['exp_avg_sq'])
            self.steps = int(state['step'])
            if self.step_dev is not None:
                self.step_dev.fill_(self.steps)
        self.lr, self.eps = float(state['lr']), float(state['eps'])
        self.betas = (float(state['betas'][0]), float(state['betas'][1]))

    def zero_grad(self, set_to_none=False):
        self.bucket.zero_grad()

    def step(self, grad_scale=1.0, gather=True):
        from .gp.module import transform_cache_active
        if transform_cache_active():
            raise RuntimeError('FusedAdam.step() inside a transform_cache scope: the raw-pointer Adam kernel does not '
                               'bump parameter versions, so the cached softplus values would go stale')
        if gather:
            self.bucket.gather_grads()
        else:
            self.bucket.check_homed()
        self.steps += 1
        if self.step_dev is not None:
            self.step_dev.add_(1)
        ops.adam_step_(self.bucket.flat_p, self.bucket.flat_g, self.exp_avg, self.exp_avg_sq, self.lr,
                       self.betas[0], self.betas[1], self.eps, self.steps, grad_scale, step_dev=self.step_dev)
