"""models.* -- the class surface of Stansfash/nonstationary-precip on the MI355X engine (nsgp).

Importing this package makes `import gpytorch` resolve to nsgp.gp when the real gpytorch is absent,
so the reference's experiments/*.py import lines work unchanged."""
import nsgp.gp as _gp

_gp.install_as_gpytorch()
