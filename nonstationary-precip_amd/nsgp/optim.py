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
    multi-tensor copy -- fewer launches per step; FusedAdam and DataParallel call it themselves."""

    def __init__(self, params, grads_as_views=True):
        seen, plist = set(), []
        for p in params:
            if p.requires_grad and id(p) not in seen:
                seen.add(id(p))
                plist.append(p)
        if not plist:
            raise ValueError('FlatBucket: no trainable parameters')
        dev, dt = plist[0].device, plist[0].dtype
        for p in plist:
            if p.device != dev or p.dtype != dt:
                raise ValueError('FlatBucket: all parameters must share device and dtype')
        self.params = plist
        self.grads_as_views = grads_as_views
        self.numel = sum(p.numel() for p in plist)
        self.flat_p = torch.empty(self.numel, dtype=dt, device=dev)
        self.flat_g = torch.zeros(self.numel, dtype=dt, device=dev)
        off = 0
        self.offsets = []
        with torch.no_grad():
            for p in plist:
                n = p.numel()
                self.flat_p[off:off + n].copy_(p.reshape(-1))
                p.data = self.flat_p[off:off + n].view(p.shape)
                p.grad = self.flat_g[off:off + n].view(p.shape) if grads_as_views else None
                self.offsets.append((off, n))
                off += n
        self.grad_views = [self.flat_g[off:off + n].view(p.shape) for p, (off, n) in zip(plist, self.offsets)]

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

    def gather_grads(self):
        """Pack the parameters' .grad tensors into the flat gradient buffer (no-op for view gradients)."""
        self.check_homed()
        if self.grads_as_views:
            return
        dst, src, missing = [], [], []
        for p, v in zip(self.params, self.grad_views):
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

    def __init__(self, params, lr=0.01, betas=(0.9, 0.999), eps=1e-8, capturable=False, grads_as_views=True):
        self.bucket = params if isinstance(params, FlatBucket) else FlatBucket(list(params), grads_as_views)
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
                'numel': int(self.bucket.numel)}

    def load_state_dict(self, state):
        """In-place restore: the moment buffers and the device step counter keep their addresses, so a hipGraph
        captured around step() stays valid."""
        if int(state['numel']) != self.bucket.numel:
            raise ValueError(f"FusedAdam.load_state_dict: {state['numel']} parameters in the checkpoint, "
                             f'{self.bucket.numel} in the bucket')
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
