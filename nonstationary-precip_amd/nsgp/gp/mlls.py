"""Objectives [gpytorch.mlls recalled, SURVEY A.5]:
ExactMarginalLogLikelihood = [log N(y | mu, K + noise I) + added-loss terms + sum priors] / N
VariationalELBO = sum_i E_q log p(y_i | f_i) / B  -  KL / num_data  (+ priors / num_data - added losses)
DeepApproximateMLL = mean over the likelihood-sample dimension
InducingPointKernelAddedLossTerm = -1/2 sum_i (k_ii - q_ii) / noise   (Titsias trace term)."""
import torch

from .. import ops
from .likelihoods import GaussianLikelihood
from .module import Module


class AddedLossTerm:
    def loss(self, *params):
        raise NotImplementedError


class InducingPointKernelAddedLossTerm(AddedLossTerm):
    def __init__(self, prior_dist, variational_dist, likelihood):
        self.prior_dist, self.variational_dist, self.likelihood = prior_dist, variational_dist, likelihood

    def loss(self, *params):
        prior_diag = self.prior_dist.lazy_covariance_matrix.diag()
        var_diag = self.variational_dist.lazy_covariance_matrix.diag()
        diag = var_diag - prior_diag
        noise_diag = self.likelihood._shaped_noise_covar(prior_diag.shape, *params).diag()
        return 0.5 * (diag / noise_diag).sum()


class MarginalLogLikelihood(Module):
    def __init__(self, likelihood, model):
        super().__init__()
        self.likelihood = likelihood
        self.model = model


class ExactMarginalLogLikelihood(MarginalLogLikelihood):
    def _add_other_terms(self, res, params):
        for term in self.model.added_loss_terms():
            res = res + term.loss(*params)
        for _, module, prior, closure, _ in self.named_priors():
            res = res + prior.log_prob(closure(module)).sum()
        return res

    def forward(self, function_dist, target, *params):
        output = self.likelihood(function_dist, *params)
        res = output.log_prob(target)
        res = self._add_other_terms(res, params)
        num_data = function_dist.event_shape.numel()
        return res / num_data


class _ApproximateMarginalLogLikelihood(MarginalLogLikelihood):
    def __init__(self, likelihood, model, num_data, beta=1.0, combine_terms=True):
        super().__init__(likelihood, model)
        self.combine_terms = combine_terms
        self.num_data = num_data
        self.beta = beta

    def _log_likelihood_term(self, approximate_dist_f, target, **kwargs):
        raise NotImplementedError

    def forward(self, approximate_dist_f, target, **kwargs):
        num_batch = approximate_dist_f.event_shape[0]
        log_likelihood = self._log_likelihood_term(approximate_dist_f, target, num_batch=num_batch, **kwargs)
        kl_divergence = self.model.variational_strategy.kl_divergence() / (self.num_data / self.beta)
        added_loss, log_prior = None, None                 # absent terms cost no launches
        for term in self.model.added_loss_terms():
            added_loss = term.loss() if added_loss is None else added_loss + term.loss()
        for _, module, prior, closure, _ in self.named_priors():
            lp = prior.log_prob(closure(module)).sum() / self.num_data
            log_prior = lp if log_prior is None else log_prior + lp
        if self.combine_terms:
            res = log_likelihood - kl_divergence
            if log_prior is not None:
                res = res + log_prior
            if added_loss is not None:
                res = res - added_loss
            return res
        if log_prior is None:
            log_prior = torch.zeros_like(log_likelihood)
        if added_loss is not None:
            return log_likelihood, kl_divergence, log_prior, added_loss
        return log_likelihood, kl_divergence, log_prior


class VariationalELBO(_ApproximateMarginalLogLikelihood):
    def _log_likelihood_term(self, variational_dist_f, target, num_batch=None, **kwargs):
        """likelihood.expected_log_prob(target, q(f)).sum(-1) / B; one fused reduction kernel for the
        Gaussian likelihood with a (S, B) diagonal q(f) (the DSVI hot path)."""
        mean, var = variational_dist_f.mean, variational_dist_f.variance
        B = num_batch if num_batch is not None else mean.shape[-1]
        if isinstance(self.likelihood, GaussianLikelihood) and mean.is_cuda and target.dim() == 1:
            shp = mean.shape[:-1]
            out = ops.GaussEllFn.apply(target, mean.reshape(-1, mean.shape[-1]), var.reshape(-1, mean.shape[-1]),
                                       self.likelihood.noise, 1.0 / B)
            return out.reshape(shp)
        return self.likelihood.expected_log_prob(target, variational_dist_f, **kwargs).sum(-1) / B


def fused_dsvi_objective(base, approximate_dist_f, target, ell_scale, kl_scale, negate=False):
    """ell_scale * sum_s sum_i E_q log p(y_i | f_si)  -  kl_scale * sum_j KL_j  as ONE scalar with a short launch
    chain (two reductions per term, device-resident upstream gradients), or None when the fast path does not
    apply (non-Gaussian likelihood, CPU tensors, priors / added-loss terms, non-whitened strategies).
    DeepApproximateMLL(VariationalELBO) is the case ell_scale = 1/(B S), kl_scale = beta/num_data; the data-parallel
    share of nsgp.dist.dp_objective uses 1/(B_global S) and beta/(num_data G)."""
    if not isinstance(base, VariationalELBO) or not base.combine_terms:
        return None
    lik = base.likelihood
    mean, var = approximate_dist_f.mean, approximate_dist_f.variance
    if not (isinstance(lik, GaussianLikelihood) and mean.is_cuda and target.dim() == 1 and mean.dim() == 2):
        return None
    strategies = getattr(base.model.variational_strategy, 'sub_variational_strategies', None)
    if strategies is None or any(True for _ in base.model.added_loss_terms()) or any(True for _ in base.named_priors()):
        return None
    pairs = []
    for st in strategies:
        vd = getattr(st, '_variational_distribution', None)
        if vd is None or not hasattr(vd, 'chol_variational_covar') or not hasattr(st, 'whiten_group'):
            return None
        pairs.append((vd.variational_mean, vd.chol_variational_covar))
    sign = -1.0 if negate else 1.0                       # negate: the loss -ELBO itself, no separate negation
    Ms = {Lq.shape[-1] for _, Lq in pairs}
    if pairs and len(Ms) == 1 and len(pairs) <= 8 and all(m.dtype == mean.dtype for m, _ in pairs):
        # every layer's KL and the likelihood term in one pair of launches (nsgp_dsvi_objective_fwd / _bwd)
        flat = [t for pr in pairs for t in pr]
        return ops.DsviObjectiveFn.apply(target, mean, var, lik.noise, sign * float(ell_scale), -sign * float(kl_scale), *flat)
    total = ops.GaussEllTotalFn.apply(target, mean, var, lik.noise, sign * float(ell_scale))
    for m, Lq in pairs:                                  # each KL term is added into the running scalar by its own kernel
        total = ops.KlWhitenedTotalFn.apply(m, Lq, -sign * float(kl_scale), total)
    return total


class DeepApproximateMLL(MarginalLogLikelihood):
    def __init__(self, base_mll):
        super().__init__(base_mll.likelihood, base_mll.model)
        self.base_mll = base_mll

    def forward(self, approximate_dist_f, target, **kwargs):
        base = self.base_mll
        if not kwargs and isinstance(base, VariationalELBO):
            mean = approximate_dist_f.mean
            if mean.dim() == 2:
                B, S = mean.shape[-1], mean.shape[0]
                fused = fused_dsvi_objective(base, approximate_dist_f, target, 1.0 / (B * S), base.beta / base.num_data)
                if fused is not None:
                    return fused
        return base(approximate_dist_f, target, **kwargs).mean(0)
