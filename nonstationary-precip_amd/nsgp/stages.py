"""Staged backward pass of a DSVI deep-GP step, so that the data-parallel gradient exchange can start on the parameters
whose gradients are already final while the rest of the backward is still running (nsgp/dist.py).

The autograd graph of a step is cut where the forward passes from one part of the model to the next:

    packed softplus of the raw hyper-parameters  |  whitening chain (Kzz, Cholesky, inverse of every layer)  |
    layer 1  |  layer 2  | ... |  last layer + ELBO

A cut replaces the tensors that cross it by detached leaves.  `BackwardStages.backward(loss)` then runs the loss's
backward (it stops at the leaves), and resolves the cuts in reverse order of creation -- each resolution is an ordinary
`torch.autograd.backward` of the cut's original tensors with the gradients its leaves collected -- calling
`after_stage(k)` in between.  No node runs twice and the parameter gradients are the ones a single backward produces
(tests/test_gpu_dist.py).  After stage k the gradients of the parameters `final_stage_of` maps to k are final: for models/dgps.py's 2-layer DeepGP
that is the last layer's variational parameters after stage 0 (4 MB of the 12.6 MB bucket, exchanged under the hidden
layer's backward), the hidden layer's after stage 1 (8 MB, exchanged under the Cholesky chain's adjoint), and the inducing
points and hyper-parameters (a few KB) at the end.

The cut sites are in nsgp/gp/models.py (DeepGP.__call__: softplus values and the whitening node's outputs;
DeepGPLayer.__call__: a layer's sampled input) and are active only inside `settings.backward_stages(plan)`."""
import torch


class BackwardStages:
    def __init__(self):
        self.cuts = []                      # [(originals, leaves)] in forward order, rebuilt by every forward pass

    def begin(self):
        """Called by DeepGP.__call__ at the start of a forward pass."""
        self.cuts = []

    def cut(self, tensors):
        """Replace `tensors` (any nesting flattened by the caller) by detached leaves; tensors that do not require grad
        pass through unchanged (and are not recorded)."""
        origs, leaves, out = [], [], []
        for t in tensors:
            if torch.is_tensor(t) and t.requires_grad:
                leaf = t.detach().requires_grad_(True)
                origs.append(t)
                leaves.append(leaf)
                out.append(leaf)
            else:
                out.append(t)
        if origs:
            self.cuts.append((origs, leaves))
        return out

    @property
    def num_stages(self):
        return len(self.cuts) + 1

    def _stage_roots(self, k, loss):
        if k == 0:
            return [loss]
        return self.cuts[len(self.cuts) - k][0]

    def run_stage(self, k, loss=None, gradient=None):
        """Stage 0: the loss's own backward (stops at the leaves of the cuts); stage k >= 1: the k-th cut from the end --
        an ordinary backward of the cut's original tensors with the gradients its leaves have collected."""
        if k == 0:
            loss.backward(gradient=gradient)
            return
        origs, leaves = self.cuts[len(self.cuts) - k]
        roots, grads = [], []
        for o, l in zip(origs, leaves):
            if l.grad is not None:
                roots.append(o)
                grads.append(l.grad)
        if roots:
            torch.autograd.backward(roots, grads)

    def backward(self, loss, gradient=None, after_stage=None):
        """All stages in order; `after_stage(k)` runs after each (e.g. DataParallel.allreduce_stage)."""
        for k in range(self.num_stages):
            self.run_stage(k, loss, gradient)
            if after_stage is not None:
                after_stage(k)

    def final_stage_of(self, loss, params):
        """{id(p): k} -- the stage after which p's gradient is final (the LAST stage whose part of the autograd graph reaches
        p; parameters the graph never reaches map to 0).  Walks the graph of one forward pass; call it once, before the
        gradient buckets are laid out (the assignment depends on the model's structure only)."""
        want = {id(p) for p in params}
        final = {pid: 0 for pid in want}
        for k in range(self.num_stages):
            seen, stack = {}, [r.grad_fn for r in self._stage_roots(k, loss) if r.grad_fn is not None]
            while stack:
                fn = stack.pop()
                if fn is None or id(fn) in seen:
                    continue
                seen[id(fn)] = fn                               # keeps the node wrapper alive: ids stay unique
                var = getattr(fn, 'variable', None)             # AccumulateGrad node of a leaf
                if var is not None and id(var) in want:
                    final[id(var)] = max(final[id(var)], k)
                stack.extend(nf for nf, _ in fn.next_functions)
        return final
