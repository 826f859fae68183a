"""A small lazy-tensor family: just the structure the reference models manipulate
(gpytorch.lazy.{delazify, LowRankRootLazyTensor, LowRankRootAddedDiagLazyTensor, DiagLazyTensor,
MatmulLazyTensor} at models/gibbs_kernels.py:191-236 and models/nonstationary_models.py:117-134).
Every dense product goes to the MFMA GEMM (nsgp.ops.matmul); nothing here touches the CPU."""
import torch

from .. import ops


def _dense(x):
    return x.evaluate() if isinstance(x, LazyTensor) else x


def delazify(x):
    return _dense(x)


def lazify(x):
    return x if isinstance(x, LazyTensor) else NonLazyTensor(x)


class LazyTensor:
    def evaluate(self):
        raise NotImplementedError

    def evaluate_kernel(self):
        return self

    @property
    def shape(self):
        return self.evaluate().shape

    def size(self, d=None):
        return self.shape if d is None else self.shape[d]

    def dim(self):
        return len(self.shape)

    @property
    def dtype(self):
        return self.evaluate().dtype

    @property
    def device(self):
        return self.evaluate().device

    @property
    def batch_shape(self):
        return self.shape[:-2]

    def diag(self):
        return torch.diagonal(self.evaluate(), dim1=-1, dim2=-2)

    def matmul(self, rhs):
        rhs = _dense(rhs)
        vec = rhs.dim() == 1
        out = ops.matmul(self.evaluate(), rhs.unsqueeze(-1) if vec else rhs)
        return out.squeeze(-1) if vec else out

    __matmul__ = matmul

    def transpose(self, a, b):
        return NonLazyTensor(self.evaluate().transpose(a, b))

    def add_diag(self, diag):
        return AddedDiagLazyTensor(self, DiagLazyTensor(diag.expand(self.shape[:-1]) if diag.dim() <= 1 else diag))

    def add_jitter(self, jitter_val=1e-3):
        n = self.shape[-1]
        d = torch.full(self.shape[:-1], float(jitter_val), dtype=self.dtype, device=self.device)
        return self.add_diag(d) if n else self

    def _mul_constant(self, c):
        return ConstantMulLazyTensor(self, c)

    def mul(self, other):
        if isinstance(other, (int, float)):
            other = torch.tensor(float(other), dtype=self.dtype, device=self.device)
        if torch.is_tensor(other) and other.numel() == 1:
            return self._mul_constant(other.reshape(()))
        if torch.is_tensor(other) and other.shape[-2:] == (1, 1):
            return ConstantMulLazyTensor(self, other)
        return NonLazyTensor(self.evaluate() * _dense(other))

    __mul__ = mul
    __rmul__ = mul

    def __add__(self, other):
        if isinstance(other, DiagLazyTensor):
            return AddedDiagLazyTensor(self, other)
        if isinstance(other, LazyTensor):
            return SumLazyTensor(self, other)
        return NonLazyTensor(self.evaluate() + other)

    __radd__ = __add__

    def __getitem__(self, idx):
        return NonLazyTensor(self.evaluate()[idx])

    def numpy(self):
        return self.evaluate().detach().cpu().numpy()

    def detach(self):
        return NonLazyTensor(self.evaluate().detach())


class NonLazyTensor(LazyTensor):
    def __init__(self, tensor):
        self.tensor = tensor

    def evaluate(self):
        return self.tensor


class LazyEvaluatedKernelTensor(LazyTensor):
    """kernel(x1, x2, **params), evaluated (once) on demand."""

    def __init__(self, x1, x2, kernel, last_dim_is_batch=False, **params):
        self.x1, self.x2, self.kernel, self.params = x1, x2, kernel, params
        self._cache = None

    @property
    def shape(self):
        b = torch.broadcast_shapes(self.x1.shape[:-2], self.x2.shape[:-2], self.kernel.batch_shape)
        return torch.Size((*b, self.x1.shape[-2], self.x2.shape[-2]))

    @property
    def dtype(self):
        return self.x1.dtype

    @property
    def device(self):
        return self.x1.device

    def evaluate_kernel(self):
        if self._cache is None:
            self._cache = lazify(self.kernel.forward(self.x1, self.x2, **self.params))
        return self._cache

    def evaluate(self):
        return self.evaluate_kernel().evaluate()

    def diag(self):
        return self.kernel(self.x1, self.x2, diag=True, **self.params)

    def _mul_constant(self, c):
        return self.evaluate_kernel()._mul_constant(c)

    def add_diag(self, diag):
        # K + c I in the same launch when the kernel can fold a constant diagonal (Gibbs build)
        if self._cache is None and torch.is_tensor(diag) and diag.numel() == 1 \
                and getattr(self.kernel, 'fuses_diag_add', False) and '_diag_add' not in self.params \
                and self.x1.shape[-2] == self.x2.shape[-2]:
            return LazyEvaluatedKernelTensor(self.x1, self.x2, self.kernel, _diag_add=diag, **self.params)
        return self.evaluate_kernel().add_diag(diag)


class DiagLazyTensor(LazyTensor):
    def __init__(self, diag):
        self._diag = diag

    @property
    def shape(self):
        return torch.Size((*self._diag.shape, self._diag.shape[-1]))

    @property
    def dtype(self):
        return self._diag.dtype

    @property
    def device(self):
        return self._diag.device

    def diag(self):
        return self._diag

    def evaluate(self):
        return torch.diag_embed(self._diag)

    def _mul_constant(self, c):
        return DiagLazyTensor(self._diag * c)

    def __add__(self, other):
        if isinstance(other, DiagLazyTensor):
            return DiagLazyTensor(self._diag + other._diag)
        return lazify(other).__add__(self) if isinstance(other, LazyTensor) else NonLazyTensor(self.evaluate() + other)


class RootLazyTensor(LazyTensor):
    """R R^T for a root R:(..., n, k)."""

    def __init__(self, root):
        self.root = lazify(root)

    @property
    def shape(self):
        r = self.root.shape
        return torch.Size((*r[:-1], r[-2]))

    @property
    def dtype(self):
        return self.root.dtype

    @property
    def device(self):
        return self.root.device

    def evaluate(self):
        R = self.root.evaluate()
        return ops.matmul(R, R, False, True)

    def diag(self):
        R = self.root.evaluate()
        return (R * R).sum(-1)

    def _mul_constant(self, c):
        # gpytorch RootLazyTensor._mul_constant: positive constants scale the root by sqrt(c)
        return self.__class__(self.root.evaluate() * torch.sqrt(c))

    def add_diag(self, diag):
        return LowRankRootAddedDiagLazyTensor(self, DiagLazyTensor(
            diag.expand(self.shape[:-1]) if diag.dim() <= 1 else diag))


class LowRankRootLazyTensor(RootLazyTensor):
    pass


class AddedDiagLazyTensor(LazyTensor):
    def __init__(self, lazy_tensor, diag_tensor):
        if isinstance(lazy_tensor, DiagLazyTensor) and not isinstance(diag_tensor, DiagLazyTensor):
            lazy_tensor, diag_tensor = diag_tensor, lazy_tensor
        self._lazy_tensor = lazify(lazy_tensor)
        self._diag_tensor = diag_tensor

    @property
    def shape(self):
        return self._lazy_tensor.shape

    @property
    def dtype(self):
        return self._lazy_tensor.dtype

    @property
    def device(self):
        return self._lazy_tensor.device

    def evaluate(self):
        K = self._lazy_tensor.evaluate()
        return K + torch.diag_embed(self._diag_tensor.diag().expand(K.shape[:-1]))

    def diag(self):
        return self._lazy_tensor.diag() + self._diag_tensor.diag()

    def _mul_constant(self, c):
        return self.__class__(self._lazy_tensor._mul_constant(c), self._diag_tensor._mul_constant(c))

    def add_diag(self, diag):
        d = diag.expand(self.shape[:-1]) if diag.dim() <= 1 else diag
        return self.__class__(self._lazy_tensor, DiagLazyTensor(self._diag_tensor.diag() + d))


class LowRankRootAddedDiagLazyTensor(AddedDiagLazyTensor):
    pass


class MatmulLazyTensor(LazyTensor):
    def __init__(self, left, right):
        self.left, self.right = lazify(left), lazify(right)

    @property
    def shape(self):
        return torch.Size((*self.left.shape[:-1], self.right.shape[-1]))

    @property
    def dtype(self):
        return self.left.dtype

    @property
    def device(self):
        return self.left.device

    def evaluate(self):
        return ops.matmul(self.left.evaluate(), self.right.evaluate())


class SumLazyTensor(LazyTensor):
    def __init__(self, *lts):
        self.lazy_tensors = [lazify(t) for t in lts]

    def evaluate(self):
        out = self.lazy_tensors[0].evaluate()
        for t in self.lazy_tensors[1:]:
            out = out + t.evaluate()
        return out

    def diag(self):
        out = self.lazy_tensors[0].diag()
        for t in self.lazy_tensors[1:]:
            out = out + t.diag()
        return out

    def _mul_constant(self, c):
        return SumLazyTensor(*[t._mul_constant(c) for t in self.lazy_tensors])


class ConstantMulLazyTensor(LazyTensor):
    def __init__(self, base, constant):
        self.base_lazy_tensor, self.constant = lazify(base), constant

    @property
    def shape(self):
        return self.base_lazy_tensor.shape

    def evaluate(self):
        return self.base_lazy_tensor.evaluate() * self.constant

    def diag(self):
        c = self.constant
        return self.base_lazy_tensor.diag() * (c.squeeze(-1) if torch.is_tensor(c) and c.dim() >= 2 else c)


class CholLazyTensor(RootLazyTensor):
    """L L^T for a lower-triangular L (CholeskyVariationalDistribution's covariance)."""

    def __init__(self, chol):
        super().__init__(chol)

    def evaluate(self):
        L = self.root.evaluate()
        return ops.matmul(L, L, False, True, a_lower=True, b_lower=True)


TriangularLazyTensor = NonLazyTensor
