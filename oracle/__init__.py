"""oracle/ -- CPU restatement of the reference hot path.  TEST INFRASTRUCTURE ONLY.

This package is the *checker* for the MI355X HIP path in ``nonstationary-precip_amd/``.
Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it.  Nothing under ``nonstationary-precip_amd/`` imports, links or executes it, and the
product fails loudly (``nsgp.BackendError``) when the HIP library is missing -- there is no CPU
fallback in the product.

What it restates (float64 torch on CPU; every function cites the reference file:line it follows,
paths relative to ``/root/reference``):

* ``functional``  -- ``utils/functional.py:14-64``  (dot, t, mv, op)
* ``dataprep``    -- ``utils/dataprep.py:9-52``     (whitening, ordered split)
* ``kernels``     -- ``models/gibbs_kernels.py:154-162`` (Gibbs), gpytorch RBF-ARD / Periodic,
                     ``models/multivariate_gibbs_kernel.py:77-150`` (Paciorek-Schervish D=2)
* ``exact``       -- ``models/gibbs_kernels.py:61-109`` (LogNormalPriorProcess),
                     ``models/nonstationary_models.py:22-62`` (DiagonalExactGP objective + predict),
                     ``models/dgps.py:113-122`` (ExactGPModel SE-ARD)
* ``sparse``      -- ``models/gibbs_kernels.py:171-266`` + ``models/nonstationary_models.py:64-153``
* ``svgp``        -- ``models/dgps.py:15-111`` on top of gpytorch's whitened VariationalStrategy /
                     DeepGPLayer / VariationalELBO / DeepApproximateMLL
* ``psgibbs``     -- ``models/latent_priors.py:27-64``, ``models/multivariate_gibbs_kernel.py:20-150``,
                     ``models/sparse_multivariate_gibbs_kernel.py:20-154``
* ``spatiotemporal`` -- ``models/spatio_temporal_models.py:17-33`` (RBF x Periodic + spatial RBF, exact and SGPR)
* ``philox``      -- Philox4x32-10 (Random123 known-answer vectors) for the partition-invariant DSVI noise

PARITY PINNING STATUS
---------------------
The arithmetic below the ``models.*`` classes lives in the third-party, un-vendored, un-pinned
``gpytorch`` (API usage implies 1.3 <= v < 1.9), which is absent from this image and cannot be
fetched; the reference ships **no tests, golden vectors or fixtures** for this path
(SURVEY.md section 8c).  Therefore:

* ``functional`` and ``dataprep`` are **pinned**: the reference's own ``utils/functional.py`` and
  ``utils/dataprep.py`` import cleanly here (torch/pandas only), and
  ``tests/golden/make_reference_goldens.py`` ran them to produce
  ``tests/golden/ref_functional.npz`` / ``ref_dataprep.npz``.
* everything that goes through gpytorch is **parity unpinned** by reference artefacts.  It is held
  instead by known-answer identities (constant-lengthscale Gibbs == RBF, SVGP at init == prior,
  Z == X optimal-q SVGP == exact GP, SGPR with Z == X == exact GP), by scikit-learn's
  ``GaussianProcessRegressor`` and scipy LAPACK for the stationary exact GP, scikit-learn's
  ``ExpSineSquared`` for the periodic kernel's functional form (gpytorch's ell-vs-ell^2 convention stays
  recalled), and by ``torch.autograd.gradcheck`` for gradients (tests/test_oracle.py).
"""

from . import functional, dataprep, kernels, exact, sparse, svgp, psgibbs, philox, spatiotemporal  # noqa: F401
