"""pulseportraiture_amd -- MI355X-native wideband pulsar-timing (TOA) fit engine.

Keeps the call surface of PulsePortraiture's pptoas hot path (`fit_portrait_full`,
`fit_portrait`, `fit_phase_shift`, `GetTOAs`, `DataBunch`) and runs it as
hand-written HIP kernels behind a C ABI (include/pp_toas.h).  There is no CPU
fallback: the HIP library must be built (see __graft_entry__.build).
"""
from .pplib import DataBunch, Dconst  # noqa: F401

__all__ = ["pplib", "pptoaslib", "pptoas", "engine"]
