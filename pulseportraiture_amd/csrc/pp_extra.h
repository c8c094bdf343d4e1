// Kernels beside the main fit: 1-D FFTFIT (fit_phase_shift) and the
// on-device synthetic portrait generator.
#pragma once
#include "pp_kernels.h"
namespace pp {
}  // namespace pp
