"""ctypes binding of libnsgp_hip.so -- the C ABI declared in include/nsgp.h.

The prototypes are parsed from the header at import time, so the binding cannot drift from the ABI
and `declared_symbols()` is what tests/test_abi.py checks the shared object against.
There is NO fallback: if the library is missing or a symbol is absent, `BackendError` is raised.
"""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(os.path.dirname(_HERE))
HEADER = os.path.join(_ROOT, 'include', 'nsgp.h')
# NSGP_LIB: developer hook for A/B timing of an experimental build of the SAME library (tools/gemm_bench.py)
LIB_PATH = os.environ.get('NSGP_LIB') or os.path.join(_HERE, 'libnsgp_hip.so')


class BackendError(RuntimeError):
    """The MI355X HIP backend is unavailable or rejected a call.  Never caught by the product."""


_CTYPES = {
    'int': ctypes.c_int, 'int64_t': ctypes.c_int64, 'int32_t': ctypes.c_int32,
    'uint64_t': ctypes.c_uint64, 'size_t': ctypes.c_size_t,
    'float': ctypes.c_float, 'double': ctypes.c_double,
}
_PROTO = re.compile(r'^\s*(int|size_t|const char\*)\s+(nsgp_\w+)\s*\(([^;{]*)\)\s*;', re.M | re.S)


def _strip_comments(text):
    text = re.sub(r'/\*.*?\*/', ' ', text, flags=re.S)
    return re.sub(r'//[^\n]*', ' ', text)


def _parse_header(path=HEADER):
    protos = {}
    text = _strip_comments(open(path).read())
    for ret, name, args in _PROTO.findall(text):
        argtypes = []
        args = ' '.join(args.split())
        if args not in ('', 'void'):
            for a in args.split(','):
                a = a.strip()
                if '*' in a:
                    argtypes.append(ctypes.c_void_p)
                else:
                    ty = a.replace('const ', '').split()[0]
                    argtypes.append(_CTYPES[ty])
        restype = {'int': ctypes.c_int, 'size_t': ctypes.c_size_t, 'const char*': ctypes.c_char_p}[ret]
        protos[name] = (restype, argtypes)
    return protos


PROTOTYPES = _parse_header()
_lib = None


def declared_symbols():
    return sorted(PROTOTYPES)


def load():
    """Load the shared object and attach the header's prototypes.  Raises BackendError."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise BackendError(
            f'{LIB_PATH} not found: build it with `python __graft_entry__.py` (hipcc --offload-arch=gfx950). '
            'nsgp has no CPU fallback.')
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as e:                                    # pragma: no cover
        raise BackendError(f'cannot load {LIB_PATH}: {e}') from e
    for name, (restype, argtypes) in PROTOTYPES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise BackendError(f'{LIB_PATH} does not export {name} declared in include/nsgp.h') from e
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def call(name, *args):
    """Call an int-returning entry point and turn a non-zero status into BackendError."""
    rc = getattr(load(), name)(*args)
    if rc != 0:
        what = f'argument {-rc} invalid' if rc < 0 else f'hipError {rc}'
        raise BackendError(f'{name} failed: {what}')
    return rc
