"""CPU-side checks of the drop-in boundary: the shared object loads and exports every symbol that
include/nsgp.h declares (no compute calls here -- there is no GPU in the build container)."""
import ctypes
import os
import re

import pytest
import torch


def test_header_declares_the_hot_path_entry_points():
    from nsgp import declared_symbols
    names = declared_symbols()
    for stem in ('gibbs_build_fwd', 'gibbs_build_bwd', 'rbf_build_fwd', 'rbf_build_bwd', 'ps2d_build_fwd',
                 'ps2d_build_bwd', 'gemm', 'potrf', 'trtri', 'svgp_colstats', 'svgp_colstats_bwd',
                 'dgp_sample_fwd', 'dgp_sample_bwd', 'gauss_ell_fwd', 'gauss_ell_bwd', 'kl_whitened_fwd',
                 'kl_whitened_bwd', 'philox_normal'):
        for sfx in ('f32', 'f64'):
            assert f'nsgp_{stem}_{sfx}' in names, (stem, sfx)
    assert 'nsgp_adam_step_f32' in names and 'nsgp_abi_version' in names


def test_library_loads_and_exports_every_declared_symbol():
    import nsgp
    assert os.path.exists(nsgp.LIB_PATH), 'run `python __graft_entry__.py` (build) first'
    lib = nsgp.load_library()                     # raises BackendError on any missing symbol
    assert lib.nsgp_abi_version() == 1
    assert lib.nsgp_build_arch() == b'gfx950'
    raw = ctypes.CDLL(nsgp.LIB_PATH)
    for name in nsgp.declared_symbols():
        assert hasattr(raw, name), name
    # host-only size queries are safe without a GPU
    assert lib.nsgp_potrf_workspace(1024, 3, 8) == 3 * 16 * 64 * 64 * 8
    assert lib.nsgp_gemm_workspace(1024, 40960, 1024, 1, 1, 4, 0) == 0
    assert lib.nsgp_gemm_workspace(1024, 1024, 40960, 1, 1, 4, 16) > 0


def test_no_signature_in_the_header_uses_torch_or_cxx_types():
    from nsgp import _lib
    text = _lib._strip_comments(open(_lib.HEADER).read())
    assert 'at::' not in text and 'torch' not in text and 'std::' not in text and 'template' not in text
    assert re.search(r'extern\s+"C"', text)


def test_product_fails_loudly_without_a_gpu_or_on_cpu_tensors():
    from nsgp import ops, BackendError
    x = torch.randn(5, 2)
    e = torch.ones(2, 5)
    with pytest.raises(BackendError):
        ops.gibbs_build(x, x, e, e)
    with pytest.raises(BackendError):
        ops.gemm(x, x, tb=True)
    with pytest.raises(BackendError):
        ops.potrf(torch.eye(4))


def test_product_never_imports_the_oracle():
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'nonstationary-precip_amd')
    for dp, _, files in os.walk(root):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', src, re.M), os.path.join(dp, f)


def test_potrf_trtri_rejects_bad_arguments_with_error_codes_not_faults():
    """nsgp_potrf_trtri_* validate their pointers and leading dimensions before any launch, like nsgp_potrf /
    nsgp_trtri (a negative return = index of the bad argument).  The checks run on the host: safe without a GPU."""
    import nsgp
    lib = nsgp.load_library()
    buf = (ctypes.c_double * 16)()
    info = (ctypes.c_int32 * 1)()
    ws = (ctypes.c_char * 65536)()
    P = lambda o: ctypes.cast(o, ctypes.c_void_p)
    ok = dict(A=P(buf), n=2, lda=2, sA=4, batch=1, info=P(info), X=P(buf), ldx=2, sX=4)

    def call(name, **kw):
        a = dict(ok, **kw)
        extra = (None, None) if name.endswith('w32') else ()
        return getattr(lib, name)(a['A'], a['n'], a['lda'], a['sA'], a['batch'], a['info'], a['X'], a['ldx'], a['sX'], *extra,
                                  P(ws), 65536, None)
    for name in ('nsgp_potrf_trtri_f32', 'nsgp_potrf_trtri_f64', 'nsgp_potrf_trtri_f64_w32'):
        assert call(name, A=None) == -1
        assert call(name, n=-1) == -2
        assert call(name, lda=1) == -3
        assert call(name, batch=-1) == -5
        assert call(name, batch=70000) == -5
        assert call(name, info=None) == -6
        assert call(name, X=None) == -7
        assert call(name, ldx=1) == -8
        assert call(name, n=0) == 0 and call(name, batch=0) == 0
