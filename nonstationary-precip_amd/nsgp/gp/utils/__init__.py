from . import cholesky, broadcasting  # noqa: F401
