"""Context-manager settings, restating the gpytorch.settings the in-scope scripts touch [SURVEY A.7]:
num_likelihood_samples (experiments/deepgp_spatial_bench.py:84), max_cg_iterations
(experiments/spatial_exp.py:199), cholesky_jitter (called as a bare no-op statement),
sgpr_diagonal_correction (models/gibbs_kernels.py:228), max_cholesky_size, debug.
Plus two knobs of this build: variational_cholesky_jitter (default 1e-4) and the
reparameterisation-noise provider used for partition-invariant data-parallel sampling."""
import torch


class _value_context:
    _global_value = None

    @classmethod
    def value(cls, *a):
        return cls._global_value

    @classmethod
    def _set_value(cls, v):
        cls._global_value = v

    def __init__(self, value):
        self._orig = self.__class__.value()
        self._new = value

    def __enter__(self):
        self.__class__._set_value(self._new)
        return self

    def __exit__(self, *exc):
        self.__class__._set_value(self._orig)
        return False


class _feature_flag:
    _state = False

    @classmethod
    def on(cls):
        return cls._state

    @classmethod
    def off(cls):
        return not cls._state

    @classmethod
    def _set_state(cls, s):
        cls._state = s

    def __init__(self, state=True):
        self.prev = self.__class__.on()
        self.state = state

    def __enter__(self):
        self.__class__._set_state(self.state)
        return self

    def __exit__(self, *exc):
        self.__class__._set_state(self.prev)
        return False


class num_likelihood_samples(_value_context):
    _global_value = 10


class max_cg_iterations(_value_context):
    """Accepted for source compatibility; this build always factors with the GPU Cholesky."""
    _global_value = 1000


class max_cholesky_size(_value_context):
    _global_value = 800


class cholesky_jitter(_value_context):
    _global_value = None

    @classmethod
    def value(cls, dtype=None):
        if cls._global_value is not None:
            return cls._global_value
        return 1e-8 if dtype == torch.float64 else 1e-6


class variational_cholesky_jitter(_value_context):
    """Jitter added to Kzz in VariationalStrategy (1e-3 in gpytorch < 1.6, settings value afterwards)."""
    _global_value = 1e-4

    @classmethod
    def value(cls, dtype=None):
        return cls._global_value


class sgpr_diagonal_correction(_feature_flag):
    _state = True


class debug(_feature_flag):
    _state = True


class fast_pred_var(_feature_flag):
    _state = False


class eps_provider(_value_context):
    """None -> torch.randn on the input's device (what Normal.rsample does in the reference);
    otherwise a callable (shape, dtype, device, call_index) -> eps, e.g. nsgp.dist.PhiloxEps."""
    _global_value = None


class chol_bwd_f64(_feature_flag):
    """Adjoint of the Kzz Cholesky inverse in float64 (on, default: what autograd does in the reference,
    where the factor and the solve are float64) or in float32 (off: 3x less MFMA time)."""
    _state = True


class forward_precision(_value_context):
    """'f32' (default): the forward projections A = W Kzx, C = Lq^T A of a float32 SVGP layer run in float32 (A accumulated
    in float64 under settings.whiten_matmul_f64).
    'bf16': BASELINE configs[4]'s "bf16 forward" where bf16 can carry it -- C = Lq^T A (O(1) operands) on
    v_mfma_f32_32x32x16_bf16 (bf16 operands, float32 accumulation; csrc/gemm_bf16.hip) from a bf16 transposed copy of A;
    A itself stays float32 / float64-accumulated.  The posterior mean is untouched, the variance carries bf16 rounding.
    'bf16_all': both projections in bf16, Kxz emitted in bf16 by the build kernel -- configs[4] to the letter.  W = L^-1 has
    entries ~1e2 whose products cancel to O(1): with 8-bit mantissas the layer outputs are off by O(1) at M = 2048
    (tests/test_gpu_bf16.py prints the measured error) -- a throughput figure only.
    Backward, Cholesky and the objective stay float32 / float64 in every mode."""
    _global_value = 'f32'


class whiten_matmul_f64(_feature_flag):
    """On (default): the whitened projection A = L^-1 Kzx of a float32 SVGP layer is accumulated in float64 (float64 MFMA
    on the float32 Kzx, rounded once) -- what the reference computes (a float64 triangular solve cast back).
    Off: one exact-float32 MFMA GEMM W Kzx, 2x the matrix-core rate on that product; its float32 accumulation of terms
    |W||Kzx| >> |A| costs ~2e-4 relative on the posterior mean at kappa(Kzz) ~ 1e6 (tools/probes/whiten_precision.py),
    above the 1e-4 parity bound.  Float64 models are unaffected."""
    _state = True


class whiten_matmul_i8(_feature_flag):
    """On: the whitened projection A = L^-1 Kzx of a float32 SVGP layer runs on the INT8 matrix cores as an exact
    digit-plane product (csrc/gemm_i8.hip: 5 signed 7-bit planes of the float64 W x 4 planes of Kzx evaluated in float64,
    14 plane products accumulated exactly in int32, combined in float64, rounded once) -- 9.5e-7 of max|A| against the
    float64 product at kappa(Kzz) ~ 1e6, where float64 accumulation of a float32 Kzx gives 5.6e-6 and float32 accumulation
    6.6e-5 -- at 64x the float64 MFMA rate per multiply-add.  Takes precedence over whiten_matmul_f64's float64 MFMA product
    and over hidden_kzx_f64 (its Kzx digits already come from a float64 evaluation); needs the float64 W of the whitening
    chain, D <= 4 input dimensions and M <= 4096.  Off: the float64-accumulating product (round 2's arithmetic)."""
    _state = True


class hidden_kzx_f64(_feature_flag):
    """On (default): a float32 layer whose output is the NEXT layer's input (a DeepGPLayer with output_dims, i.e. every layer
    of a deep GP but the last) builds its Kzx in float64 and feeds it to the float64-accumulating projection
    (nsgp_svgp_tri_gemm_colstats_f64acc_b64); everything downstream of A stays float32.  Why: the float32 rounding of Kzx,
    amplified by |W||Kzx| ~ 1e2, is 3e-5 of such a layer's mean, and a trained next layer multiplies it ten-fold -- after
    1000 Adam steps of the headline model the output mean is 3.8e-4 off the float64 result (the reference's own float32
    arithmetic: 4.5e-4), with this switch 2.6e-5 (tools/probes/precision_after_training.py).  Costs one float64 kernel build
    per hidden layer (~+0.03 ms per step at the headline shape).  Off: the reference's arithmetic (float32 Kzx everywhere).
    Needs whiten_matmul_f64 (the float64 W); float64 models are unaffected."""
    _state = True


class hidden_var_f64(_value_context):
    """With hidden_kzx_f64: also run the SECOND projection C = Lq^T A of such a layer on the float64-accumulating kernel, with
    float64 column-statistic partials, so that its variance os + colsum(C^2 - A^2) -- a difference of two O(os) sums that
    cancels to << os once q(u) has trained -- carries no float32 partial-sum rounding (4e-8 absolute = 5e-5 of a small
    variance, which reaches the next layer through sqrt(var) eps: output mean 5.4e-5 instead of 2.5e-5 after 1000 steps).
    True / False, or 'auto' (default): on where the layer sees at most 8192 points per output GP -- the first hidden layer
    of a deep GP (its inputs are the minibatch); deeper hidden layers see S x minibatch points, where the float64 product
    costs as much as the whole float32 layer (+12 % on a BASELINE configs[4] step)."""
    _global_value = 'auto'


class check_mvn_cholesky(_feature_flag):
    """On (default): MultivariateNormal.log_prob reads the Cholesky `info` of its dense covariance (one host sync) and
    follows psd_safe_cholesky -- jitter retries with a NumericalWarning, then NotPSDError -- instead of returning a NaN
    objective.  Off: sync-free (for graph capture of an exact-GP step; skipped automatically while a stream captures)."""
    _state = True


class check_variational_cholesky(_feature_flag):
    """Off (default): the DSVI step never synchronises with the host -- a Kzz that is not positive definite shows up as
    NaNs in the ELBO (gpytorch would have raised from psd_safe_cholesky after its jitter retries).  On: read the
    factorisation's LAPACK-style `info` after every whitening chain (one host sync per model call) and raise
    NotPSDError naming the failing GP and leading minor; meant for debugging, not for graph-captured training."""
    _state = False


class backward_stages(_value_context):
    """An nsgp.stages.BackwardStages plan: while set, DeepGP / DeepGPLayer cut the autograd graph between the parts of
    the model so that the backward pass can be run part by part (gradient exchange overlapped with it, nsgp/dist.py).
    None (default): one ordinary backward."""
    _global_value = None



class fuse_kzx(_feature_flag):
    """On: a float32 SVGP layer's forward projection A = W Kzx generates the Kzx tiles inside the GEMM loader
    (nsgp_svgp_kzx_gemm_colstats_f64acc) instead of reading a materialised Kzx -- same values bit for bit, one (M x n)
    matrix less through HBM per layer and step (the backward still builds it once for Wbar = tril(Abar Kzx^T)).
    Off (default): Kzx is built by the RBF kernel first.  Measured at the headline shape: the generator (a row of z per
    K-tile through scalar loads, a division per coordinate, four exp) slows the product from 1.08 to 1.23 ms per step,
    while the (M x n) round trip it saves was never the bound: 5.10 ms/step with this off, 5.28 with it on.  Kept as a
    memory-saving mode (168 MB at n = 40960)."""
    _state = False
