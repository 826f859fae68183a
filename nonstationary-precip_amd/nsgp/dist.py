"""Data-parallel DSVI over RCCL/xGMI: one process per GPU, parameters replicated, the minibatch
sharded, ONE sum-all-reduce of the flat gradient bucket per step.

    loss_r = -( (B_r / B) * mean_s[ sum_{i in r} ELL_{s,i} / B_r ] - KL / (G * N_data) )
    sum_r loss_r == the single-GPU loss   (so the all-reduce is a plain SUM, no rescale)

The reparameterisation noise comes from a counter-based generator keyed by the GLOBAL row index,
so the union of the ranks' draws equals the single-GPU draw (partition invariance).
Kzz build + Cholesky are parameter-only work and are replicated, not sharded.
`backend='nccl'` is RCCL on ROCm; the CPU tests use gloo."""
import torch
import torch.distributed as dist

from . import ops


def shard_bounds(n, world, rank):
    """Contiguous split of n rows over `world` ranks (first n % world ranks get one more)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class PhiloxEps:
    """settings.eps_provider callable: eps[s, i, c] keyed by (seed, step, call index, global row).
    With `step_dev` (device int64, e.g. FusedAdam(capturable=True).step_dev) the step is read on the
    device, so a captured hipGraph draws fresh noise at every replay."""

    def __init__(self, seed, row0=0, step_dev=None):
        self.seed, self.row0 = int(seed), int(row0)
        self.step, self.call = 0, 0
        self.step_dev = step_dev

    def start_step(self, step, row0=None):
        self.step, self.call = int(step), 0
        if row0 is not None:
            self.row0 = int(row0)

    def __call__(self, shape, dtype, device):
        S, n, b = shape
        stream_id = (self.step << 32) | self.call
        self.call += 1
        return ops.philox_normal(self.seed, stream_id, self.row0, S, n, b, dtype=dtype, device=device,
                                 step_dev=self.step_dev)


def dp_objective(mll, output, y_local, batch_global, world, negate=False):
    """Rank-local share of DeepApproximateMLL(VariationalELBO(...))(output, y): see module docstring.
    negate=True returns the loss (minus the objective) directly, sparing the negation kernels."""
    base = getattr(mll, 'base_mll', mll)
    b_local = y_local.shape[-1]
    mean = output.mean
    if mean.dim() == 2:
        from .gp.mlls import fused_dsvi_objective
        fused = fused_dsvi_objective(base, output, y_local, 1.0 / (batch_global * mean.shape[0]),
                                     base.beta / (base.num_data * world), negate=negate)
        if fused is not None:
            return fused
    ell = base._log_likelihood_term(output, y_local, num_batch=b_local)            # (S,)
    kl = base.model.variational_strategy.kl_divergence() / (base.num_data / base.beta)
    obj = ((b_local / batch_global) * ell - kl / world).mean(0)
    return -obj if negate else obj


class DataParallel:
    """Wraps a FlatBucket: `allreduce_grads()` sums the bucket over the process group."""

    def __init__(self, bucket, group=None, force=False):
        """force=True: issue the collectives even in a one-rank group (rehearsal of the N>1 call sequence on one GPU)."""
        self.bucket, self.group = bucket, group
        self._pending = []
        self.stages_issued, self.stages_waited = 0, 0        # all-reduces started / joined (check_drained)
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.active = self.world > 1 or (force and dist.is_initialized())

    def broadcast_params(self, src=0):
        if self.active:
            dist.broadcast(self.bucket.flat_p, src=src, group=self.group)

    def allreduce_stage(self, stage, gather=True):
        """Staged backward (nsgp/stages.py): start the sum-all-reduce of the gradients that backward stage `stage`
        completed -- one contiguous range of the bucket -- without waiting for it; `wait_stages()` joins them all.  The
        collective runs on the process group's own stream, so kernels launched afterwards (the next stage of the
        backward) overlap with it.  Same SUM over the same values as `allreduce_grads`, in up to `num_stages` pieces."""
        if gather:
            self.bucket.gather_grads(stage)
        seg = self.bucket.segments.get(stage)
        if self.active and seg is not None and seg[1] > seg[0]:
            self._pending.append(dist.all_reduce(self.bucket.flat_g[seg[0]:seg[1]], op=dist.ReduceOp.SUM,
                                                 group=self.group, async_op=True))
            self.stages_issued += 1

    def wait_stages(self):
        """Join every all-reduce `allreduce_stage` started (an exception of a collective propagates to the caller)."""
        pending, self._pending = self._pending, []
        for work in pending:
            work.wait()                       # device tensors: the current stream waits, the host does not block
            self.stages_waited += 1

    def check_drained(self):
        """Raise unless every exchange that was started has been waited on: the optimiser step must not read a bucket
        range whose all-reduce is still in flight (call right before the Adam step)."""
        if self._pending or self.stages_issued != self.stages_waited:
            raise RuntimeError(f'DataParallel: {len(self._pending)} gradient all-reduce(s) still pending '
                               f'({self.stages_issued} issued, {self.stages_waited} waited): call wait_stages() before '
                               'the optimiser step')

    def allreduce_grads(self, async_op=False, gather=True):
        if gather:
            self.bucket.gather_grads()
        if self.active:
            return dist.all_reduce(self.bucket.flat_g, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)
        return None
