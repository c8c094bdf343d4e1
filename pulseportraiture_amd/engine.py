"""Thin Python wrapper of the C ABI: one Engine = one device context.

The batch entry point `Engine.fit_batch` is what reaches 10^4 fits/s; the
reference-shaped single-subint functions in pptoaslib.py / pplib.py marshal
into it.
"""
import ctypes as C
import os
import threading

import numpy as np

from . import _lib
from ._lib import (FitIn, FitOut, PP_ENOTSUP, PP_F32, PP_F64, PP_METHOD_NEWTON, PP_METHOD_TRUST_NCG,
                   SeedRef, c_double_p, c_int32_p, c_uint8_p)

# the reference's minimiser names (pptoaslib.py:993-1010) -> device solver
METHODS = {'trust-ncg': PP_METHOD_TRUST_NCG, 'Newton-CG': PP_METHOD_NEWTON,
           'TNC': PP_METHOD_NEWTON, 'newton': PP_METHOD_NEWTON}


class EngineError(RuntimeError):
    pass


class EngineNotSupported(EngineError):
    """The library has no device path for this request in this shape (PP_ENOTSUP): nothing
    was done; the caller takes its general route."""


def _check(rc, what):
    if rc == PP_ENOTSUP:
        raise EngineNotSupported("%s: %s" % (what, _lib.last_error()))
    if rc != 0:
        raise EngineError("%s failed (%d): %s" % (what, rc, _lib.last_error()))


def _is_device_array(x):
    """A device-resident array (torch tensor).  The engine launches on its OWN HIP
    stream, the array's contents were produced on the caller's (torch's current
    stream, asynchronously): wait for the producer before the engine may touch it --
    otherwise a kernel of ours can read a tensor whose `clone()` / `to()` / fill has
    not finished (seen as rare, run-dependent garbage in a few rows).  The engine's
    calls end with a synchronise of its own stream, so the other direction is safe."""
    if hasattr(x, "data_ptr") and hasattr(x, "is_cuda") and bool(x.is_cuda):
        _wait_for_producer(x)
        return True
    return False


# Devices whose producer stream the engine call running on THIS thread has already waited for.
# Per thread: engines are one per host thread, and a set shared between threads would let one
# thread's call clear -- or pre-fill -- the record of another's while that one sits in
# synchronize() with the GIL released.
_TLS = threading.local()


def _new_call():
    """Start of an engine call: every device array handed over was produced before it, so
    one wait per device and call covers them all (seven waits per fit cost ~50 us)."""
    _TLS.waited = set()


def _wait_for_producer(x):
    waited = getattr(_TLS, "waited", None)
    if waited is None:
        waited = _TLS.waited = set()
    key = str(x.device)
    if key in waited:
        return
    try:
        import torch
        torch.cuda.current_stream(x.device).synchronize()
    except ImportError:      # (another array library: its arrays must be complete when handed over)
        pass
    waited.add(key)


def _dp(a):
    return None if a is None else a.ctypes.data_as(c_double_p)


def _f64(a, shape=None):
    if a is None:
        return None
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None:
        a = np.ascontiguousarray(np.broadcast_to(a, shape))
    return a


def _nu_array(nus, nsub):
    """[nsub,3] float64 reference frequencies; None entries become NaN (= let
    the engine choose: mean frequency for the fit, zero-covariance for output)."""
    if nus is None:
        return np.full((nsub, 3), np.nan)
    a = np.asarray(nus)
    if a.dtype == object:
        a = np.where(np.equal(a, None), np.nan, a).astype(np.float64)
    return np.ascontiguousarray(np.broadcast_to(a.astype(np.float64, copy=False),
                                                (nsub, 3)))


class Engine(object):
    """Device context + scratch; not thread-safe (one per host thread and GPU)."""

    def __init__(self, device=0):
        self._lib = _lib.load()
        ctx = C.c_void_p()
        _check(self._lib.pp_create(int(device), C.byref(ctx)), "pp_create")
        self._ctx = ctx
        self.device = int(device)
        if os.environ.get("PP_DEBUG_POISON"):
            # work buffers start every batch filled with NaN bit patterns: a kernel that reads
            # what no kernel of the batch wrote shows up in the results (a poor man's
            # sanitizer; run the GPU tests once with PP_DEBUG_POISON=255)
            self.set_option("debug_poison", int(os.environ["PP_DEBUG_POISON"]))
        if os.environ.get("PP_OVERLAP_POST"):
            # (run the GPU suite once with the solve / post-fit stage of enqueued batches on the context's second
            # stream: PP_OVERLAP_POST=1)
            self.set_option("overlap_post", int(os.environ["PP_OVERLAP_POST"]))
        self._digests = {}      # slot -> content digest of the resident template

    def close(self):
        if getattr(self, "_ctx", None):
            self._lib.pp_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- configuration --------------------------------------------------
    def set_option(self, name, value):
        _check(self._lib.pp_set_option(self._ctx, name.encode(), float(value)),
               "pp_set_option(%s)" % name)
        if name == "harm_eps":
            self._digests.clear()      # truncation is decided when a template is set

    def get_option(self, name):
        """Current value of an engine option (pp_get_option)."""
        v = C.c_double()
        _check(self._lib.pp_get_option(self._ctx, name.encode(), C.byref(v)), "pp_get_option(%s)" % name)
        return v.value

    def synchronize(self):
        _check(self._lib.pp_synchronize(self._ctx), "pp_synchronize")

    # -- model ------------------------------------------------------------
    def set_model(self, model, slot=0, digest=None):
        """Upload an nchan x nbin template (numpy array or CUDA tensor).
        `digest` (optional) tags the slot's content for set_model_cached."""
        self._digests[int(slot)] = digest
        if _is_device_array(model):
            nchan, nbin = int(model.shape[-2]), int(model.shape[-1])
            dtype = PP_F64 if model.element_size() == 8 else PP_F32
            ptr, on_dev, keep = model.data_ptr(), 1, model
        else:
            m = np.asarray(model)
            if m.dtype != np.float32:
                m = m.astype(np.float64, copy=False)
            m = np.ascontiguousarray(m)
            nchan, nbin = m.shape
            dtype = PP_F64 if m.dtype == np.float64 else PP_F32
            ptr, on_dev, keep = m.ctypes.data, 0, m
        _check(self._lib.pp_model_set(self._ctx, int(slot), C.c_void_p(ptr), dtype,
                                      on_dev, nchan, nbin), "pp_model_set")
        del keep
        return self._lib.pp_model_nharm(self._ctx, int(slot))

    def set_model_cached(self, model, slot=0):
        """Upload the template unless the same bytes are already resident in the
        slot (the reference re-reads and re-transforms it for every subint,
        pptoas.py:352-379)."""
        m = np.ascontiguousarray(model)
        try:
            import xxhash
            digest = (m.shape, m.dtype.str,
                      xxhash.xxh3_128_digest(m.view(np.uint8).reshape(-1)))
        except ImportError:
            digest = None
        if digest is None or self._digests.get(int(slot)) != digest:
            self.set_model(m, slot=slot, digest=digest)
        return self.model_nharm(slot)

    def model_nharm(self, slot=0):
        return self._lib.pp_model_nharm(self._ctx, int(slot))

    # -- the fit ------------------------------------------------------------
    def fit_batch(self, data, freqs, P, init_params, errs=None, nu_fits=None,
                  nu_outs=None, fit_flags=(1, 1, 0, 0, 0), log10_tau=False,
                  option=0, is_toa=True, model_slot=None, chan_mask=None,
                  per_channel=True, objective=False, seed_ns=0, method='trust-ncg',
                  records=None, ref_seed=None, _submit=False):
        """Fit nsub subints.  data: [nsub,nchan,nbin] numpy array (f64/f32) or
        CUDA tensor.  freqs: [nchan] or [nsub,nchan].  Returns a dict of arrays
        (see include/pp_toas.h pp_fit_out).  errs / chan_mask may be CUDA
        tensors [nsub,nchan]; per_channel="device" leaves scales, scale_errs and
        channel_snrs in HBM as CUDA tensors instead of copying them out;
        records (a float64 CUDA tensor [nsub, 18]) receives one TOA record per
        subint on the device (dist.RECORD_FIELDS).
        seed_ns > 0 replaces init_params[:, 0] by a phase seeded on the device.
        method: 'trust-ncg' follows SciPy's trust-ncg iteration to the very point
        where the reference stops; 'newton' ('Newton-CG', 'TNC') converges to the
        rounding of the objective in fewer evaluations.
        ref_seed: dict(weights=[nsub,nchan] (array, CUDA tensor or None), model_profs=[nsub,nbin]
        or [nbin], nu_mean=[nsub], Ns=100, bounds=(-0.5, 0.5), finish='simplex') -- the
        reference's own phase guess (pptoas.py:421-457) is formed inside the fit, from the same
        single pass over the portraits, and replaces init_params[:, 0]; the result gains
        "seed_phase".  Raises EngineNotSupported when the batch has no single-pass path
        (include/pp_toas.h pp_seed_ref): form the guess with reference_phase_seed then."""
        if method not in METHODS:
            raise EngineError("unknown method %r" % (method,))
        if records is not None and not (
                _is_device_array(records) and records.is_contiguous() and
                records.element_size() == 8 and tuple(records.shape)[-1] == _lib.PP_RECORD_WIDTH):
            raise EngineError("records must be a contiguous float64 CUDA tensor [nsub, %d]"
                              % _lib.PP_RECORD_WIDTH)
        if _is_device_array(data):
            nsub, nchan, nbin = (int(s) for s in data.shape)
            if not data.is_contiguous():
                raise EngineError("device data must be contiguous")
            dtype = PP_F64 if data.element_size() == 8 else PP_F32
            dptr, on_dev, keep = data.data_ptr(), 1, data
        else:
            d = np.asarray(data)
            if d.dtype != np.float32:
                d = d.astype(np.float64, copy=False)
            d = np.ascontiguousarray(d)
            if d.ndim == 2:
                d = d[None]
            nsub, nchan, nbin = d.shape
            dtype = PP_F64 if d.dtype == np.float64 else PP_F32
            dptr, on_dev, keep = d.ctypes.data, 0, d
        freqs = np.ascontiguousarray(freqs, dtype=np.float64)
        if freqs.ndim == 1:
            if freqs.shape[0] != nchan:
                raise EngineError("freqs has %d entries for %d channels" %
                                  (freqs.shape[0], nchan))
            fstride = 0
        else:
            if freqs.shape != (nsub, nchan):
                raise EngineError("freqs shape %s != (%d, %d)" %
                                  (freqs.shape, nsub, nchan))
            fstride = nchan
        P = _f64(P, (nsub,))
        x0 = _f64(init_params, (nsub, 5))
        aux_dev = _is_device_array(errs) or _is_device_array(chan_mask)
        if aux_dev:
            for t, nm in ((errs, "errs"), (chan_mask, "chan_mask")):
                if t is not None and not (_is_device_array(t) and t.is_contiguous()
                                          and tuple(t.shape) == (nsub, nchan)):
                    raise EngineError("%s must be a contiguous CUDA tensor "
                                      "[nsub,nchan] when either aux input is" % nm)
            if errs is not None and errs.element_size() != 8:
                raise EngineError("device errs must be float64")
            if chan_mask is not None and chan_mask.element_size() != 1:
                raise EngineError("device chan_mask must be uint8")
        else:
            errs = _f64(errs, (nsub, nchan))
        nu_fits = _nu_array(nu_fits, nsub)
        nu_outs = _nu_array(nu_outs, nsub)
        slot = None if model_slot is None else np.ascontiguousarray(
            np.broadcast_to(model_slot, (nsub,)), dtype=np.int32)
        mask = None
        if chan_mask is not None and not aux_dev:
            mask = np.ascontiguousarray(np.broadcast_to(chan_mask, (nsub, nchan)),
                                        dtype=np.uint8)

        fin = FitIn()
        fin.nsub, fin.nchan, fin.nbin = nsub, nchan, nbin
        fin.data = C.c_void_p(dptr)
        fin.data_dtype, fin.data_on_device = dtype, on_dev
        fin.model_slot = None if slot is None else slot.ctypes.data_as(c_int32_p)
        fin.freqs, fin.freqs_stride = _dp(freqs), fstride
        fin.aux_on_device = int(aux_dev)
        if aux_dev:
            fin.errs = None if errs is None else C.cast(errs.data_ptr(), c_double_p)
            fin.chan_mask = None if chan_mask is None else C.cast(
                chan_mask.data_ptr(), c_uint8_p)
        else:
            fin.errs = _dp(errs)
            fin.chan_mask = None if mask is None else mask.ctypes.data_as(c_uint8_p)
        fin.P, fin.init_params = _dp(P), _dp(x0)
        fin.nu_fits, fin.nu_outs = _dp(nu_fits), _dp(nu_outs)
        for j in range(5):
            fin.fit_flags[j] = 1 if fit_flags[j] else 0
        fin.log10_tau = int(bool(log10_tau))
        fin.option, fin.is_toa = int(option), int(bool(is_toa))
        fin.seed_ns = int(seed_ns)
        fin.method = METHODS[method]
        seed_keep = None
        if ref_seed is not None:
            sr = SeedRef()
            w = ref_seed.get("weights")
            if w is not None and _is_device_array(w):
                if not aux_dev and (errs is not None or chan_mask is not None):
                    raise EngineError("ref_seed weights on the device need errs / chan_mask there too")
                if not (w.is_contiguous() and w.element_size() == 8 and tuple(w.shape) == (nsub, nchan)):
                    raise EngineError("device ref_seed weights must be a contiguous float64 tensor [nsub,nchan]")
                fin.aux_on_device = 1
                sr.weights = C.cast(w.data_ptr(), c_double_p)
            elif w is not None:
                if aux_dev:
                    raise EngineError("ref_seed weights must be a CUDA tensor when errs / chan_mask are")
                w = _f64(w, (nsub, nchan))
                sr.weights = _dp(w)
            mp = np.ascontiguousarray(ref_seed["model_profs"], dtype=np.float64)
            if mp.ndim == 1:
                sr.model_prof_stride = 0
            elif mp.shape == (nsub, nbin):
                sr.model_prof_stride = nbin
            else:
                raise EngineError("ref_seed model_profs must be [nbin] or [nsub,nbin]")
            if mp.shape[-1] != nbin:
                raise EngineError("ref_seed model_profs has %d bins, the data %d" % (mp.shape[-1], nbin))
            sr.model_profs = _dp(mp)
            numean = _f64(ref_seed["nu_mean"], (nsub,))
            sr.nu_mean = _dp(numean)
            lo_hi = ref_seed.get("bounds", (-0.5, 0.5))
            sr.lo, sr.hi = float(lo_hi[0]), float(lo_hi[1])
            sr.Ns = int(ref_seed.get("Ns", 100))
            sr.finish = 1 if ref_seed.get("finish", "simplex") == "simplex" else 0
            sphase = np.empty(nsub)
            sr.seed_phase = _dp(sphase)
            fin.ref_seed = C.pointer(sr)
            seed_keep = (sr, w, mp, numean, sphase)

        res = dict(params=np.empty((nsub, 5)), param_errs=np.empty((nsub, 5)),
                   nu_refs=np.empty((nsub, 3)), cov=np.empty((nsub, 5, 5)),
                   chi2=np.empty(nsub), red_chi2=np.empty(nsub), snr=np.empty(nsub),
                   nfeval=np.empty(nsub, dtype=np.int32),
                   return_code=np.empty(nsub, dtype=np.int32),
                   npass=np.empty(nsub, dtype=np.int32),
                   duration=np.zeros(1))
        chan_dev = (per_channel == "device")
        if chan_dev:
            import torch
            dev = data.device if on_dev else torch.device("cuda", self.device)
            res.update({k: torch.empty((nsub, nchan), dtype=torch.float64, device=dev)
                        for k in ("scales", "scale_errs", "channel_snrs")})
        elif per_channel:
            res.update(scales=np.empty((nsub, nchan)),
                       scale_errs=np.empty((nsub, nchan)),
                       channel_snrs=np.empty((nsub, nchan)))
        if objective:
            res.update(obj_f=np.empty(nsub), obj_grad=np.empty((nsub, 5)),
                       obj_hess=np.empty((nsub, 5, 5)))
        fout = FitOut()
        fout.chan_on_device = int(chan_dev)
        if records is not None:
            if int(records.shape[0]) != nsub:
                raise EngineError("records has %d rows for %d subints" % (records.shape[0], nsub))
            fout.records_dev = C.cast(records.data_ptr(), c_double_p)
        for name, _ in FitOut._fields_:
            if name in ("chan_on_device", "records_dev"):
                continue
            arr = res.get(name)
            if arr is None:
                setattr(fout, name, None)
            elif _is_device_array(arr):
                setattr(fout, name, C.cast(arr.data_ptr(), c_double_p))
            elif arr.dtype == np.int32:
                setattr(fout, name, arr.ctypes.data_as(c_int32_p))
            else:
                setattr(fout, name, arr.ctypes.data_as(c_double_p))
        res["fit_flags"] = [1 if f else 0 for f in fit_flags]
        if seed_keep is not None:
            res["seed_phase"] = seed_keep[4]
        if _submit == "enqueue":
            _check(self._lib.pp_fit_enqueue(self._ctx, C.byref(fin), C.byref(fout)), "pp_fit_enqueue")
            # every array the argument blocks point to stays alive until collect()
            if not hasattr(self, "_queue"):
                self._queue = []
            self._queue.append((res, (keep, freqs, P, x0, errs, nu_fits, nu_outs, slot, mask, chan_mask,
                                      records, fin, fout, seed_keep)))
            return None
        if _submit:
            _check(self._lib.pp_fit_submit(self._ctx, C.byref(fin), C.byref(fout)), "pp_fit_submit")
            # every array the argument blocks point to stays alive until wait()
            self._pending = (res, (keep, freqs, P, x0, errs, nu_fits, nu_outs, slot, mask, chan_mask,
                                   records, fin, fout, seed_keep))
            return None
        _check(self._lib.pp_fit_portrait_batch(self._ctx, C.byref(fin),
                                               C.byref(fout)),
               "pp_fit_portrait_batch")
        del keep, seed_keep
        res["duration"] = float(res["duration"][0])
        return res

    def submit(self, *args, **kwargs):
        """fit_batch started on a worker thread of the context (pp_fit_submit): returns at
        once; wait() returns the result dict.  The caller's arrays must not be modified
        until then.  Two engines on one GPU overlap their copies and kernels."""
        kwargs["_submit"] = True
        self.fit_batch(*args, **kwargs)

    def enqueue(self, *args, **kwargs):
        """fit_batch queued on the engine's stream (pp_fit_enqueue): returns without waiting for the
        GPU; collect() returns the result dict of the OLDEST enqueued batch.  Up to three batches may be
        pending: enqueue batch k + 1, then collect batch k, and the GPU never waits for the host
        between batches.  The caller's arrays must not be modified until the batch is collected."""
        kwargs["_submit"] = "enqueue"
        self.fit_batch(*args, **kwargs)

    def collect(self):
        """Complete the oldest enqueued batch; returns what fit_batch returns."""
        if not getattr(self, "_queue", None):
            raise EngineError("nothing enqueued")
        # the entry (result arrays + keep-alive references of everything the argument blocks point to) leaves the
        # queue only once the C call is through with it; if that call fails, the Python queue is brought back in
        # step with the library's (pp_fit_pending): the failed batch is gone there, younger ones may still be queued
        res, keep = self._queue[0]
        try:
            _check(self._lib.pp_fit_collect(self._ctx), "pp_fit_collect")
        finally:
            left = self._lib.pp_fit_pending(self._ctx)
            left = left if left >= 0 else 0
            while len(self._queue) > left:
                self._queue.pop(0)
        del keep
        res["duration"] = float(res["duration"][0])
        return res

    def poll(self):
        """True once the submitted batch is complete."""
        rc = self._lib.pp_fit_poll(self._ctx)
        if rc < 0:
            _check(rc, "pp_fit_poll")
        return bool(rc)

    def wait(self):
        """Block until the submitted batch is complete; returns what fit_batch returns."""
        if getattr(self, "_pending", None) is None:
            raise EngineError("nothing submitted")
        res, keep = self._pending
        self._pending = None
        _check(self._lib.pp_fit_wait(self._ctx), "pp_fit_wait")
        del keep
        res["duration"] = float(res["duration"][0])
        return res

    # -- parity hooks / measurement ---------------------------------------
    def rfft_rows(self, rows):
        r = np.asarray(rows)
        if r.dtype != np.float32:
            r = r.astype(np.float64, copy=False)
        r = np.ascontiguousarray(r)
        nrows, nbin = r.shape
        out = np.empty((nrows, nbin // 2 + 1), dtype=np.complex128)
        _check(self._lib.pp_rfft_rows(self._ctx, C.c_void_p(r.ctypes.data),
                                      PP_F64 if r.dtype == np.float64 else PP_F32,
                                      nrows, nbin,
                                      out.ctypes.data_as(c_double_p)),
               "pp_rfft_rows")
        return out

    def fit_phase_shift_batch(self, data, model, noise=None, bounds=(-0.5, 0.5),
                              Ns=100, finish='newton'):
        """1-D FFTFIT of nprof profile pairs: Ns-point brute grid over `bounds` (both
        ends included), then finish = 'newton': the exact local optimum; 'simplex':
        what scipy.optimize.brute does by default and the reference returns
        (pplib.py:2085) -- Nelder-Mead to xtol = ftol = 1e-4, step for step.
        Returns [nprof, 7]: phase, phase_err, scale, scale_err, snr, red_chi2, duration."""
        if finish not in ('newton', 'simplex'):
            raise ValueError("finish must be 'newton' or 'simplex'")
        d = _f64(np.atleast_2d(data))
        nprof, nbin = d.shape
        m = _f64(np.atleast_2d(model), (nprof, nbin))
        if noise is None:
            nz = np.full(nprof, np.nan)
        elif isinstance(noise, np.ndarray) and noise.dtype.kind == "f":
            nz = _f64(noise, (nprof,))
        else:
            nz = _f64([np.nan if v is None else v for v in
                       np.broadcast_to(np.asarray(noise, dtype=object), (nprof,))])
        out = np.empty((nprof, 7))
        self.set_option("fps_finish", 1 if finish == 'simplex' else 0)
        try:
            _check(self._lib.pp_fit_phase_shift_batch(
                self._ctx, _dp(d), _dp(m), _dp(nz), nprof, nbin, float(bounds[0]),
                float(bounds[1]), int(Ns), _dp(out)), "pp_fit_phase_shift_batch")
        finally:
            self.set_option("fps_finish", 0)
        return out

    def reference_phase_seed(self, ports, freqs, P, weights, model_profs, phi=0.0, DM=0.0, GM=0.0,
                             nu_DM=np.inf, nu_GM=np.inf, bounds=(-0.5, 0.5), Ns=100, finish='simplex'):
        """fit_phase_shift(np.average(rotate_data(port_i, phi_i, DM_i, P_i, freqs_i, nu_DM),
        axis=0, weights=weights_i), model_profs_i, Ns) for every subint, the rotation
        and the channel mean fused into one read of the portraits (pptoas.py:421-457).
        Returns [nsub, 7] like fit_phase_shift_batch."""
        src, dtype, on_dev, (nsub, nchan, nbin), keep = self._ports_arg(ports)
        freqs = np.ascontiguousarray(freqs, dtype=np.float64)
        fstride = 0 if freqs.ndim == 1 else nchan
        P = _f64(np.broadcast_to(np.asarray(P, dtype=np.float64), (nsub,)))
        par = np.ascontiguousarray(np.stack([
            np.broadcast_to(np.asarray(v, dtype=np.float64), (nsub,)) for v in (phi, DM, GM)], axis=1))
        w = _f64(weights, (nsub, nchan))
        mp = _f64(np.broadcast_to(np.asarray(model_profs, dtype=np.float64), (nsub, nbin)))
        out = np.empty((nsub, 7))
        self.set_option("fps_finish", 1 if finish == 'simplex' else 0)
        try:
            _check(self._lib.pp_reference_phase_seed(
                self._ctx, src, dtype, on_dev, nsub, nchan, nbin, _dp(freqs), fstride, _dp(P), _dp(par),
                float(nu_DM), float(nu_GM), _dp(w), _dp(mp), float(bounds[0]), float(bounds[1]), int(Ns),
                _dp(out)), "pp_reference_phase_seed")
        finally:
            self.set_option("fps_finish", 0)
        return out

    def rotate_portraits(self, ports, freqs, P, phi=0.0, DM=0.0, GM=0.0, nu_DM=np.inf,
                         nu_GM=np.inf):
        """Fourier-rotate ports[nsub,nchan,nbin] (numpy -> new numpy array; CUDA
        tensor -> rotated in place) by per-subint (phi, DM, GM)."""
        if _is_device_array(ports):
            nsub, nchan, nbin = (int(v) for v in ports.shape)
            dtype = PP_F64 if ports.element_size() == 8 else PP_F32
            src = dst = C.c_void_p(ports.data_ptr())
            on_dev, out = 1, ports
        else:
            a = np.asarray(ports)
            if a.dtype != np.float32:
                a = a.astype(np.float64, copy=False)
            a = np.ascontiguousarray(a)
            nsub, nchan, nbin = a.shape
            dtype = PP_F64 if a.dtype == np.float64 else PP_F32
            out = np.empty_like(a)
            src, dst, on_dev = C.c_void_p(a.ctypes.data), C.c_void_p(out.ctypes.data), 0
        freqs = np.ascontiguousarray(freqs, dtype=np.float64)
        fstride = 0 if freqs.ndim == 1 else nchan
        P = _f64(P, (nsub,))
        par = np.ascontiguousarray(np.stack([
            np.broadcast_to(np.asarray(v, dtype=np.float64), (nsub,)) for v in (phi, DM, GM)],
            axis=1))
        _check(self._lib.pp_rotate_portraits(self._ctx, src, dst, dtype, on_dev, nsub, nchan,
                                             nbin, _dp(freqs), fstride, _dp(P), _dp(par),
                                             float(nu_DM), float(nu_GM)),
               "pp_rotate_portraits")
        return out

    @staticmethod
    def _gauss_args(model, P):
        p = np.asarray(model["params"], dtype=np.float64)
        ng = int(model["ngauss"])
        comps = np.ascontiguousarray(p[2:2 + 6 * ng].reshape(ng, 6))
        tau_rot = 0.0
        if p[1] != 0.0:
            if P is None:
                raise ValueError("need the period P for a model with TAU != 0")
            tau_rot = float(p[1]) / float(P)
        return (str(model["code"]).encode(), float(model["nu_ref"]), float(p[0]), tau_rot,
                float(model["alpha"]), ng, comps)

    def gaussian_portrait(self, model, freqs, nbin, P=None, out=None):
        """nchan x nbin portrait of a parsed .gmodel (gmodel.parse_gmodel) built
        on the device; `out` may be a CUDA float64 tensor [nchan,nbin] (filled in
        place), else a NumPy array is returned."""
        freqs = _f64(freqs)
        nchan = len(freqs)
        code, nu_ref, dc, tau_rot, alpha, ng, comps = self._gauss_args(model, P)
        if out is not None and _is_device_array(out):
            dst, on_dev, ret = C.c_void_p(out.data_ptr()), 1, out
        else:
            ret = np.empty((nchan, int(nbin)))
            dst, on_dev = C.c_void_p(ret.ctypes.data), 0
        _check(self._lib.pp_gaussian_portrait(self._ctx, nchan, int(nbin), _dp(freqs), code, nu_ref,
                                              dc, tau_rot, alpha, ng, _dp(comps), dst, on_dev),
               "pp_gaussian_portrait")
        return ret

    def set_model_gaussian(self, model, freqs, nbin, P=None, slot=0):
        """Synthesise a .gmodel template on the device straight into a model slot;
        returns the number of harmonics kept."""
        freqs = _f64(freqs)
        code, nu_ref, dc, tau_rot, alpha, ng, comps = self._gauss_args(model, P)
        _check(self._lib.pp_model_set_gaussian(self._ctx, int(slot), len(freqs), int(nbin),
                                               _dp(freqs), code, nu_ref, dc, tau_rot, alpha, ng,
                                               _dp(comps)), "pp_model_set_gaussian")
        self._digests.pop(int(slot), None)
        return int(self._lib.pp_model_nharm(self._ctx, int(slot)))

    def _spline_args(self, mean_prof, eigvec, tck, nbin):
        from .splmodel import spline_device_args
        basis, t, coefs, k = spline_device_args(mean_prof, eigvec, tck, nbin)
        ncomp = basis.shape[0] - 1
        return basis, t, coefs, k, ncomp

    def spline_portrait(self, mean_prof, eigvec, tck, freqs, nbin=None):
        """nchan x nbin portrait of a spline model (gen_spline_portrait, pplib.py:932-956)
        built on the device; returns a NumPy array."""
        freqs = _f64(freqs)
        basis, t, coefs, k, ncomp = self._spline_args(mean_prof, eigvec, tck, nbin)
        out = np.empty((len(freqs), basis.shape[1]))
        _check(self._lib.pp_spline_portrait(self._ctx, len(freqs), basis.shape[1], _dp(freqs), ncomp,
                                            _dp(basis), len(t), _dp(t), _dp(coefs), k,
                                            C.c_void_p(out.ctypes.data), 0), "pp_spline_portrait")
        return out

    def set_model_spline(self, mean_prof, eigvec, tck, freqs, nbin=None, slot=0):
        """Synthesise a spline (.spl) template on the device straight into a model
        slot (no host portrait); returns the number of harmonics kept."""
        freqs = _f64(freqs)
        basis, t, coefs, k, ncomp = self._spline_args(mean_prof, eigvec, tck, nbin)
        _check(self._lib.pp_model_set_spline(self._ctx, int(slot), len(freqs), basis.shape[1], _dp(freqs),
                                             ncomp, _dp(basis), len(t), _dp(t), _dp(coefs), k),
               "pp_model_set_spline")
        self._digests.pop(int(slot), None)
        return int(self._lib.pp_model_nharm(self._ctx, int(slot)))

    def apply_response(self, slot, rconst=None, smear_wid=None):
        """Multiply the template resident in `slot` by an instrumental response in the
        Fourier domain, on the device: rconst[nbin/2 + 1] (complex, the product of the
        constant responses) and/or smear_wid[nchan] (dispersive smearing width of each
        channel in rotations).  pptoaslib.py:145-179."""
        rc = None if rconst is None else np.ascontiguousarray(
            np.asarray(rconst, dtype=np.complex128)).view(np.float64)
        wd = None if smear_wid is None else _f64(smear_wid)
        _check(self._lib.pp_model_apply_response(self._ctx, int(slot), _dp(rc), _dp(wd)),
               "pp_model_apply_response")
        self._digests.pop(int(slot), None)
        return int(self._lib.pp_model_nharm(self._ctx, int(slot)))

    def model_means(self, slot, nchan, nbin):
        """Mean over phase of every channel of the template in `slot` (its DC
        harmonic / nbin)."""
        dc = np.empty(int(nchan))
        _check(self._lib.pp_model_dc(self._ctx, int(slot), _dp(dc)), "pp_model_dc")
        return dc / float(nbin)

    def _ports_arg(self, ports):
        if _is_device_array(ports):
            nsub, nchan, nbin = (int(v) for v in ports.shape)
            dtype = PP_F64 if ports.element_size() == 8 else PP_F32
            return C.c_void_p(ports.data_ptr()), dtype, 1, (nsub, nchan, nbin), ports
        a = np.asarray(ports)
        if a.dtype != np.float32:
            a = a.astype(np.float64, copy=False)
        a = np.ascontiguousarray(a)
        return (C.c_void_p(a.ctypes.data), PP_F64 if a.dtype == np.float64 else PP_F32, 0,
                a.shape, a)

    def align_accumulate(self, ports, freqs, P, phase, DM, nu_ref, weights):
        """ppalign's accumulation (ppalign.py:199-206): returns
        (sum_i w[i,n] * rotate_data(ports[i,n], phase_i, DM_i, P_i, freqs, nu_ref_i)
        as [nchan,nbin], sum_i w[i,n] as [nchan]); rows with w = 0 are skipped."""
        src, dtype, on_dev, (nsub, nchan, nbin), keep = self._ports_arg(ports)
        freqs = np.ascontiguousarray(freqs, dtype=np.float64)
        fstride = 0 if freqs.ndim == 1 else nchan
        P = _f64(P, (nsub,))
        par = np.ascontiguousarray(np.stack([
            np.broadcast_to(np.asarray(v, dtype=np.float64), (nsub,))
            for v in (phase, DM, nu_ref)], axis=1))
        w = _f64(weights, (nsub, nchan))
        aligned = np.empty((nchan, nbin))
        totw = np.empty(nchan)
        _check(self._lib.pp_align_accumulate(self._ctx, src, dtype, on_dev, nsub, nchan, nbin,
                                             _dp(freqs), fstride, _dp(P), _dp(par), _dp(w),
                                             _dp(aligned), _dp(totw)), "pp_align_accumulate")
        return aligned, totw

    def channel_red_chi2(self, ports, freqs, P, params, nu_refs, scales, errs, slots=None):
        """Per-channel reduced chi^2 of fitted subints, time domain, dof = nbin-2
        (get_channels_to_zap, pptoas.py:1239-1245).  params[nsub,5] = phi, DM, GM,
        tau [rot, linear], alpha at nu_refs[nsub,3]; the template is the resident
        model slot of each subint."""
        src, dtype, on_dev, (nsub, nchan, nbin), keep = self._ports_arg(ports)
        freqs = np.ascontiguousarray(freqs, dtype=np.float64)
        fstride = 0 if freqs.ndim == 1 else nchan
        P = _f64(P, (nsub,))
        params = _f64(params, (nsub, 5))
        nu_refs = _f64(nu_refs, (nsub, 3))
        scales = _f64(scales, (nsub, nchan))
        errs = _f64(errs, (nsub, nchan))
        sl = None
        if slots is not None:
            sl = np.ascontiguousarray(np.broadcast_to(np.asarray(slots, dtype=np.int32), (nsub,)))
        out = np.empty((nsub, nchan))
        _check(self._lib.pp_channel_red_chi2(
            self._ctx, src, dtype, on_dev, nsub, nchan, nbin,
            None if sl is None else sl.ctypes.data_as(c_int32_p), _dp(freqs), fstride, _dp(P),
            _dp(params), _dp(nu_refs), _dp(scales), _dp(errs), _dp(out)), "pp_channel_red_chi2")
        return out

    def synth_portraits(self, dst, freqs, P, inj, sigma, seed, first_subint=0,
                        slot=0, gains=None):
        """Fill a CUDA tensor dst[nsub,nchan,nbin] with synthetic subints; gains[nsub,nchan]
        (optional) scales the template in every channel (scintillation)."""
        if not _is_device_array(dst):
            raise EngineError("synth_portraits needs a CUDA tensor destination")
        nsub = int(dst.shape[0])
        freqs = _f64(freqs)
        P = _f64(P, (nsub,))
        inj = _f64(inj, (nsub, 3))
        gains = _f64(gains, (nsub, int(dst.shape[1])))
        _check(self._lib.pp_synth_portraits(
            self._ctx, int(slot), C.c_void_p(dst.data_ptr()),
            PP_F64 if dst.element_size() == 8 else PP_F32, nsub, _dp(freqs),
            _dp(P), _dp(inj), _dp(gains), float(sigma), C.c_uint64(int(seed)),
            C.c_int64(int(first_subint))), "pp_synth_portraits")

    def kernel_times(self, reset=False):
        n = 16
        names = (C.c_char_p * n)()
        secs = (C.c_double * n)()
        cnt = (C.c_int64 * n)()
        k = self._lib.pp_kernel_times(self._ctx, n, names, secs, cnt)
        out = {names[i].decode(): (secs[i], cnt[i]) for i in range(k)}
        if reset:
            self._lib.pp_kernel_times_reset(self._ctx)
        return out


_default = {}


def default_engine(device=0):
    """Process-wide engine for the reference-shaped single-call API."""
    eng = _default.get(device)
    if eng is None:
        eng = _default[device] = Engine(device)
    return eng


_WHILE_PENDING = ("poll", "wait", "close")      # what may be called while a submitted batch runs
_WHILE_QUEUED = ("enqueue", "fit_batch", "collect", "close", "set_option", "get_option", "kernel_times", "synchronize")


def _fresh_waits(fn):
    import functools

    @functools.wraps(fn)
    def call(self, *args, **kwargs):
        # a submitted batch owns the context (its buffers, stream and counters) until wait():
        # include/pp_toas.h forbids every other call meanwhile -- enforce it here
        if getattr(self, "_pending", None) is not None and fn.__name__ not in _WHILE_PENDING:
            raise EngineError("a submitted batch is pending on this engine: wait() before %s()" % fn.__name__)
        if getattr(self, "_queue", None) and fn.__name__ not in _WHILE_QUEUED:
            raise EngineError("enqueued batches are pending on this engine: collect() before %s()" % fn.__name__)
        _new_call()
        return fn(self, *args, **kwargs)
    return call


# every entry point that may be handed device arrays starts a new round of producer waits
for _name in [n for n, f in vars(Engine).items() if callable(f) and not n.startswith("_")]:
    setattr(Engine, _name, _fresh_waits(getattr(Engine, _name)))

