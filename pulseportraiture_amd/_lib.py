"""ctypes binding of libpptoas_hip.so (the C ABI of include/pp_toas.h).

There is no CPU fallback: if the HIP library has not been built, importing the
engine raises.  Build with `python -c "import __graft_entry__ as g; g.build()"`
or `make -C pulseportraiture_amd/csrc`.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# PP_TOAS_LIB: another build of the same library (kernel experiments, trace builds under variants/);
# like the default it must exist and carry the binding's ABI -- there is no fallback either way
LIB_PATH = os.environ.get("PP_TOAS_LIB") or os.path.join(_HERE, "csrc", "libpptoas_hip.so")

PP_OK, PP_EINVAL, PP_EHIP, PP_ENOMEM, PP_ESTATE, PP_ENOTSUP = 0, -1, -2, -3, -4, -5
PP_F64, PP_F32 = 0, 1
PP_MAX_SLOTS = 64
PP_RECORD_WIDTH = 18
ABI_VERSION = 5
PP_METHOD_TRUST_NCG, PP_METHOD_NEWTON = 0, 1

c_double_p = C.POINTER(C.c_double)
c_int32_p = C.POINTER(C.c_int32)
c_uint8_p = C.POINTER(C.c_uint8)


class SeedRef(C.Structure):
    _fields_ = [("weights", c_double_p), ("model_profs", c_double_p), ("model_prof_stride", C.c_int64),
                ("nu_mean", c_double_p), ("lo", C.c_double), ("hi", C.c_double), ("Ns", C.c_int32),
                ("finish", C.c_int32), ("seed_phase", c_double_p)]


class FitIn(C.Structure):
    _fields_ = [("nsub", C.c_int32), ("nchan", C.c_int32), ("nbin", C.c_int32),
                ("data", C.c_void_p), ("data_dtype", C.c_int32),
                ("data_on_device", C.c_int32), ("model_slot", c_int32_p),
                ("freqs", c_double_p), ("freqs_stride", C.c_int64),
                ("errs", c_double_p), ("chan_mask", c_uint8_p),
                ("aux_on_device", C.c_int32), ("P", c_double_p),
                ("init_params", c_double_p), ("nu_fits", c_double_p),
                ("nu_outs", c_double_p), ("fit_flags", C.c_int32 * 5),
                ("log10_tau", C.c_int32), ("option", C.c_int32),
                ("is_toa", C.c_int32), ("method", C.c_int32), ("seed_ns", C.c_int32),
                ("ref_seed", C.POINTER(SeedRef))]


class FitOut(C.Structure):
    _fields_ = [("params", c_double_p), ("param_errs", c_double_p),
                ("nu_refs", c_double_p), ("cov", c_double_p), ("chi2", c_double_p),
                ("red_chi2", c_double_p), ("snr", c_double_p),
                ("nfeval", c_int32_p), ("return_code", c_int32_p),
                ("chan_on_device", C.c_int32), ("scales", c_double_p), ("scale_errs", c_double_p),
                ("channel_snrs", c_double_p), ("obj_f", c_double_p),
                ("obj_grad", c_double_p), ("obj_hess", c_double_p),
                ("duration", c_double_p), ("npass", c_int32_p), ("records_dev", c_double_p)]


# every symbol include/pp_toas.h declares: (restype, argtypes)
SYMBOLS = {
    "pp_abi_version": (C.c_int, []),
    "pp_last_error": (C.c_char_p, []),
    "pp_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "pp_destroy": (C.c_int, [C.c_void_p]),
    "pp_synchronize": (C.c_int, [C.c_void_p]),
    "pp_stream": (C.c_void_p, [C.c_void_p]),
    "pp_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_double]),
    "pp_get_option": (C.c_int, [C.c_void_p, C.c_char_p, c_double_p]),
    "pp_model_set": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int,
                               C.c_int, C.c_int]),
    "pp_model_nharm": (C.c_int, [C.c_void_p, C.c_int]),
    "pp_model_dc": (C.c_int, [C.c_void_p, C.c_int, c_double_p]),
    "pp_fit_portrait_batch": (C.c_int, [C.c_void_p, C.POINTER(FitIn),
                                        C.POINTER(FitOut)]),
    "pp_fit_submit": (C.c_int, [C.c_void_p, C.POINTER(FitIn), C.POINTER(FitOut)]),
    "pp_fit_poll": (C.c_int, [C.c_void_p]),
    "pp_fit_enqueue": (C.c_int, [C.c_void_p, C.POINTER(FitIn), C.POINTER(FitOut)]),
    "pp_fit_collect": (C.c_int, [C.c_void_p]),
    "pp_fit_pending": (C.c_int, [C.c_void_p]),
    "pp_fit_wait": (C.c_int, [C.c_void_p]),
    "pp_rfft_rows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                               c_double_p]),
    "pp_fit_phase_shift_batch": (C.c_int, [C.c_void_p, c_double_p, c_double_p,
                                           c_double_p, C.c_int, C.c_int,
                                           C.c_double, C.c_double, C.c_int,
                                           c_double_p]),
    "pp_reference_phase_seed": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                          C.c_int, c_double_p, C.c_int64, c_double_p, c_double_p,
                                          C.c_double, C.c_double, c_double_p, c_double_p, C.c_double,
                                          C.c_double, C.c_int, c_double_p]),
    "pp_rotate_portraits": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                      C.c_int, C.c_int, C.c_int, C.c_int, c_double_p,
                                      C.c_int64, c_double_p, c_double_p, C.c_double,
                                      C.c_double]),
    "pp_synth_portraits": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                     C.c_int, c_double_p, c_double_p, c_double_p, c_double_p,
                                     C.c_double, C.c_uint64, C.c_int64]),
    "pp_gaussian_portrait": (C.c_int, [C.c_void_p, C.c_int, C.c_int, c_double_p, C.c_char_p,
                                       C.c_double, C.c_double, C.c_double, C.c_double, C.c_int,
                                       c_double_p, C.c_void_p, C.c_int]),
    "pp_model_set_gaussian": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, c_double_p,
                                        C.c_char_p, C.c_double, C.c_double, C.c_double,
                                        C.c_double, C.c_int, c_double_p]),
    "pp_spline_portrait": (C.c_int, [C.c_void_p, C.c_int, C.c_int, c_double_p, C.c_int, c_double_p,
                                     C.c_int, c_double_p, c_double_p, C.c_int, C.c_void_p, C.c_int]),
    "pp_model_set_spline": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, c_double_p, C.c_int,
                                      c_double_p, C.c_int, c_double_p, c_double_p, C.c_int]),
    "pp_model_apply_response": (C.c_int, [C.c_void_p, C.c_int, c_double_p, c_double_p]),
    "pp_align_accumulate": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                      C.c_int, C.c_int, c_double_p, C.c_int64, c_double_p,
                                      c_double_p, c_double_p, c_double_p, c_double_p]),
    "pp_channel_red_chi2": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                      C.c_int, C.c_int, C.POINTER(C.c_int32), c_double_p,
                                      C.c_int64, c_double_p, c_double_p, c_double_p,
                                      c_double_p, c_double_p, c_double_p]),
    "pp_kernel_times": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_char_p),
                                  c_double_p, C.POINTER(C.c_int64)]),
    "pp_kernel_times_reset": (C.c_int, [C.c_void_p]),
}

_lib = None


class HipLibraryMissing(RuntimeError):
    pass


def load():
    """Load the HIP library (once).  Raises HipLibraryMissing if it is absent:
    the product path never falls back to a CPU implementation."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipLibraryMissing(
            "%s not found: build it with `make -C %s` (hipcc, gfx950). "
            "pulseportraiture_amd has no CPU fallback." %
            (LIB_PATH, os.path.dirname(LIB_PATH)))
    # torch ships its own libamdhip64 (same soname as the system one): when
    # torch is installed it must be the first to load the HIP runtime, or its
    # later lazy initialisation finds "No HIP GPUs" in a process where the
    # system runtime is already resident.  Importing it here makes the load
    # order deterministic; the library itself has no torch dependency.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)      # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    if lib.pp_abi_version() != ABI_VERSION:
        raise RuntimeError("libpptoas_hip.so ABI %d != binding ABI %d" %
                           (lib.pp_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


def last_error():
    return load().pp_last_error().decode("utf-8", "replace")
