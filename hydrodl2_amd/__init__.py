"""hydrodl2_amd -- MI355X-native HBV forward/backward time-stepper.

Drop-in for the model-plugin API of mhpi/hydrodl2 for ONE hot path: the HBV
per-day recurrence, its parameter prep, ensemble mean and unit-hydrograph
routing, forward and adjoint, as hand-written HIP kernels for gfx950 behind the
C ABI of include/hbvx.h.  See DESIGN.md.
"""
from hydrodl2_amd.api import available_models, available_modules, load_model, load_module

__version__ = "0.1.0"

__all__ = ["__version__", "available_models", "available_modules", "load_model", "load_module"]
