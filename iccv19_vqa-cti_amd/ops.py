"""Tensor-level wrappers over the C ABI: argument checking, output allocation (torch = device memory + stream
plumbing only) and the launch.  Every function requires CUDA(ROCm) tensors and raises otherwise -- there is no
CPU or eager-PyTorch path in this package."""
import torch

from . import _lib as L
from . import _lib as L_      # (functions that use L as a local size keep the module reachable)

_PREC = {"fp32": L.PREC_F32, "bf16x3": L.PREC_BF16X3, "bf16": L.PREC_BF16, "f16f6": L.PREC_F16F6}
# Default: 'f16f6' = the mode bench.py's headline line runs.  Outside the fused TCNet.forward / TriAttention.forward with more than 6 answer
# tokens it IS bf16x3 (fp32-grade, 1e-5 vs float64 truth at the BASELINE shapes, 3/16 of the exact-fp32 MFMA cost); inside, the a side and the
# mode-3 product run on f16 + block-scaled fp6 planes (3e-5) under the range guard below, which re-runs a call as bf16x3 when an operand leaves
# the format's domain -- so the default never trades range or NaN semantics for speed (a call nobody can wait for -- hipGraph replay -- is NaN-filled
# instead, except on the mild cancellation bit; set_range_check).
_default_prec = "f16f6"


def set_precision(name):
    """'fp32' (exact fp32 MFMA), 'bf16x3' (3-term split-bf16, fp32-grade), 'f16f6' (the default: fused TCNet.forward on f16 + block-scaled fp6
    products, fp32-grade INSIDE the format's domain -- magnitudes in f16's normal range, 6e-5 ... 65504, finite -- and guarded outside it: the
    call is re-run as bf16x3, see set_range_check; everything else as bf16x3) or 'bf16'.

    A-priori accuracy of the mode-3 product  out = sum_k M_k A^_k  (K = 512 at BASELINE configs[1]), as absolute error per output:
        fp32 (the reference, src/Tensor.py:16-20)   ~2^-24 sum_k |M_k A^_k|
        bf16x3                                      ~2^-19 sum_k |M_k A^_k|   (dropped lo x lo term + bf16 roundings of the lo parts; 2^-17 worst case)
        f16f6                                       ~2^-17 sum_k |M_k A^_k|   (each cross term rounded to the 4 bits of e2m3; 2^-15 worst case)
    so the error NORMALISED by the largest output grows with rho = sum_k |M_k A^_k| / max |out|.  Measured on the whole forward at the configs[1]
    widths (tests/test_accuracy_envelope_gpu.py; the operands M and A^ carry their own rounding, which a cancelling sum amplifies just the same):
    at most ~1.8e-5 rho as f16f6 (2.7e-5 on the synthetic tensors, rho 1.4-1.8), ~0.9e-5 rho as bf16x3, ~3e-7 rho in fp32.  The guard estimates rho
    on the device from 1 024 (row, answer) pairs per sample -- per operand 16 evenly spaced rows and the largest row of each of 16 strata, so a few
    outsized answer tokens cannot hide; numerator and denominator are maxima over the whole call, as the tolerance is -- and re-runs the call as
    bf16x3 when it exceeds 2.75, as exact fp32 when it exceeds 5.5 (set_cancel_thresholds; include/cti_hip.h: CTI_GUARD_CANCEL / _HEAVY): a factor
    two under 1e-4 on either law.  The estimate is a sample, not a bound."""
    global _default_prec
    if name not in _PREC:
        raise ValueError("precision must be one of %s" % sorted(_PREC))
    _default_prec = name


def get_precision():
    return _default_prec


# ---- range guard of the f16f6 forward (include/cti_hip.h: cti_tcnet_forward_guard_bytes / cti_guard_read) -----------------------------
# The reference multiplies in full-range fp32 (src/Tensor.py:12,18); the f16f6 operand format is fp32-grade only inside f16's normal range.
# The library scans every encoded operand before the mode-3 product and NaN-fills the output when one leaves the domain.  'sync' (default):
# the wrapper waits for that verdict -- it arrives while the mode-3 product is still running, so the launch pipeline does not drain -- and
# re-runs the call in the bf16x3 mode: the caller always gets fp32-grade numbers.  'poison': no host wait (what hipGraph capture forces): an
# out-of-range call returns NaN, never clamped numbers.
_range_check = __import__("os").environ.get("CTI_RANGE_CHECK", "sync")      # (the environment variable: A/B of the host wait)
_range_log = {"calls": 0, "trips": 0, "last_status": 0, "consecutive": 0, "skip": 0, "last_ratio": None}
_range_debug = False       # tests / diagnostics: also read the cancellation estimate rho of every guarded call (one more small host read) into last_ratio
_guard_res = {}


def set_range_check(mode):
    """'sync' (default): the wrapper waits for the guard's verdict and re-runs an out-of-domain call (the caller always gets fp32-grade numbers).
    'poison' (also what hipGraph capture forces, whatever this is set to): no host wait -- a call whose operands leave the f16f6 format's RANGE
    (status bits 1, 2, 4: saturation / non-finite, underflow) or whose cancellation estimate rho is beyond the HEAVY threshold (bit 16: rho > 5.5, and rho has
    no upper bound there) returns all-NaN output, never clamped or inaccurate numbers.  Bit 8 alone (2.75 < rho <= 5.5) leaves the f16f6 result in place: by
    the measured law (error <= ~1.8e-5 rho of the largest output, tests/test_accuracy_envelope_gpu.py) that result is still within 1e-4, and an all-NaN batch
    under a replayed graph has no remedy.  f16f6_device_status() reads the status word of the last guarded call -- e.g. after a replay --
    and graph.GraphedForward does so after every replay and re-runs a tripped call eagerly as bf16x3 / fp32.  set_poison_bits(31) NaN-fills on every bit,
    set_poison_bits(7) on the range bits only."""
    global _range_check
    if mode not in ("sync", "poison"):
        raise ValueError("range check must be 'sync' or 'poison'")
    _range_check = mode


_poison_bits_nohost = 23


def set_poison_bits(bits):
    """Status bits that NaN-fill the output of a guarded call nobody waits for ('poison' mode / hipGraph capture).  Default 23 = the range bits (1, 2, 4) and
    the HEAVY cancellation bit (16); bit 8 -- a mild cancellation estimate, error still <= ~1e-4 -- keeps its result."""
    global _poison_bits_nohost
    bits = int(bits)
    if not 0 <= bits <= 31:
        raise ValueError("poison bits: a mask of the five status bits")
    _poison_bits_nohost = bits


def set_cancel_thresholds(bf16x3=2.75, fp32=5.5):
    """Cancellation estimate rho beyond which a guarded f16f6 call is re-run as bf16x3 / as exact fp32 (this host thread's calls; the library's
    defaults are 2.75 / 5.5: a factor two under 1e-4 on the measured error laws ~1.8e-5 rho and ~0.9e-5 rho)."""
    lib = L.lib()
    L.check(lib.cti_set_tuning(L.TUNE_GUARD_RHO_BF16X3, int(round(float(bf16x3) * 1000))), "cti_set_tuning")
    L.check(lib.cti_set_tuning(L.TUNE_GUARD_RHO_FP32, int(round(float(fp32) * 1000))), "cti_set_tuning")


# The guard block (first 256 B of a guarded call's workspace) of the LAST guarded call per device, copied stream-ordered into a small tensor that lives as long
# as the process (a reference to the workspace itself pinned GBs and kept the caching allocator from recycling the block).  Under graph capture the
# copy is a node of the graph, so after a replay the tensor holds what THAT replay found.
_guard_status_block = {}


def _keep_guard_block(ws):
    key = ws.device.index if ws.device.index is not None else torch.cuda.current_device()
    blk = _guard_status_block.get(key)
    if blk is None:
        if torch.cuda.is_current_stream_capturing():
            return                        # (memory born under capture belongs to that graph's pool; an eager guarded call -- any warm-up -- creates the block)
        blk = _guard_status_block[key] = torch.zeros(256, device=ws.device, dtype=torch.uint8)
    blk.copy_(ws[:256], non_blocking=True)


def f16f6_device_status(device=None, sync=True):
    """Status word (and rho) of the LAST guarded f16f6 call on `device`, read from the copy of its guard block -- the way to learn what a replayed hipGraph's
    forward found.  sync=False: the caller has already waited for the stream the call ran on.  None when no guarded call has run."""
    key = torch.cuda.current_device() if device is None else (device.index if isinstance(device, torch.device) and device.index is not None else
                                                              (device if isinstance(device, int) else torch.cuda.current_device()))
    blk = _guard_status_block.get(key)
    if blk is None:
        return None
    if sync:
        torch.cuda.synchronize(blk.device)
    w = blk[:128].cpu().view(torch.int32)
    f = lambda i: float(w[i:i + 1].view(torch.float32)[0])
    return {"status": int(w[0]) & 0xffffffff, "rho": f(2), "abs_max": f(16), "dot_max": f(17)}


def f16f6_range_status():
    """Counters of the f16f6 range guard in this process: guarded calls, trips (calls re-run as bf16x3 / fp32), the last status word
    (bit 0 saturation / non-finite in an encoded operand, bit 1 underflow, bit 2 non-finite V^ / Q^ / T_eff, bit 3 / 4 heavy cancellation in the
    mode-3 product: re-run as bf16x3 / as exact fp32), last_ratio = the cancellation estimate rho (only with ops._range_debug = True)."""
    return dict(_range_log)


_range_owner = {}          # (device, T_g storage) -> {"consecutive", "skip", "skip_mode"}: the repeat-offender shortcut is per TCNet, not per process


def _owner_state(device, T_g):
    key = (device.index if device.index is not None else torch.cuda.current_device(), T_g.data_ptr())
    st = _range_owner.get(key)
    if st is None:
        if len(_range_owner) > 256:
            _range_owner.clear()
        st = _range_owner[key] = {"consecutive": 0, "skip": 0, "skip_mode": "bf16x3"}
    return st


def _guard_resources(device):
    # one event pair + stream per device AND host thread: two threads driving one device must not wait on each other's records
    key = (device.index if device.index is not None else torch.cuda.current_device(), __import__("threading").get_ident())
    r = _guard_res.get(key)
    if r is None:
        lib = L.lib()
        r = _guard_res[key] = (lib.cti_event_create(), lib.cti_event_create(), torch.cuda.Stream(device=device), lib.cti_event_create())
    return r


def _prec(p, fused=False):
    """Precision code of a launch.  'f16f6' exists for the fused TCNet.forward (cti_tcnet_forward / cti_tcnet_prepare) only: every other op runs its
    bf16x3 form in that mode (same fp32-grade accuracy class)."""
    c = _PREC[p or _default_prec]
    return L.PREC_BF16X3 if (c == L.PREC_F16F6 and not fused) else c


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_device = getattr(torch._C, "_cuda_getDevice", None)


def _stream():
    """The current stream's hipStream_t as an integer.  torch.cuda.current_stream() builds a Python Stream object per call (~8 us; a training
    step asks ~190 times); the raw query is what torch's own extensions use."""
    if _raw_stream is not None and _cur_device is not None:
        return _raw_stream(_cur_device())
    return torch.cuda.current_stream().cuda_stream


# Modules cache what they derive from their parameters (weight-norm scales, packed rank nets, the prepared block of the fused forward,
# concatenated glimpse projections), keyed by each parameter's storage pointer and autograd version counter.  An update that goes
# around the version counter -- FlatAdamaxDP's fused kernel writes the flat parameter buffer directly; user code doing
# `p.data.mul_(...)` -- must call invalidate_caches(), which every cache key also contains.
_param_epoch = [0]


def invalidate_caches():
    _param_epoch[0] += 1


import os as _os


class tuning:
    """Context manager over cti_set_tuning (tests / benchmarks only): `with ops.tuning(gemm_cfg=2, tri_chunk=64): ...` forces the plane GEMM's
    tile geometry (0 = 128x128, 1 = 256x128, 2 = 256x256) and / or the Tri softmax's chunk length, so small tensors run the instantiations that
    BASELINE configs[1] selects; gemm16_sk = 1 cuts every plannable product of cti_gemm_bf16_rows_sk stream-K (default: only where it pays).  Restores the
    previous values on exit."""

    def __init__(self, gemm_cfg=None, tri_chunk=None, gemm16_sk=None):
        self.want = {L.TUNE_GEMM_CFG: gemm_cfg, L.TUNE_TRI_CHUNK: tri_chunk, L.TUNE_GEMM16_SK: gemm16_sk}

    def __enter__(self):
        lib = L.lib()
        self.old = {k: lib.cti_get_tuning(k) for k, v in self.want.items() if v is not None}
        for k, v in self.want.items():
            if v is not None:
                L.check(lib.cti_set_tuning(k, int(v)), "cti_set_tuning")
        return self

    def __exit__(self, *exc):
        lib = L.lib()
        for k, v in self.old.items():
            L.check(lib.cti_set_tuning(k, int(v)), "cti_set_tuning")
        return False


_aux = {}
_AUX_PRIORITY = int(_os.environ.get("CTI_AUX_PRIORITY", "0"))      # -1 = the auxiliary chain's workgroups are dispatched ahead of the caller's stream (A/B knob)
use_aux_stream = _os.environ.get("CTI_NO_AUX_STREAM", "0") != "1"


def _aux_stream(device):
    """One side stream per device for cti_tcnet_forward's second chain (None disables the overlap).  Buffers touched on it are
    allocated on the current stream and only reused after the call's join event, so the caching allocator stays consistent."""
    o = aux_stream_object(device)
    return None if o is None else o.cuda_stream


def aux_stream_object(device, which=0):
    """The torch.cuda.Stream behind _aux_stream (one per device), or None when the overlap is disabled.  which = 1: a second side stream of the model
    forwards (the hoisted q / a projections beside the attention, whose library call occupies the first one)."""
    if not use_aux_stream or _no_nested_fork[0]:
        return None
    key = (device.index if device.index is not None else torch.cuda.current_device(), which)
    if key not in _aux:
        _aux[key] = torch.cuda.Stream(device=device, priority=_AUX_PRIORITY)
    return _aux[key]


_sib = {}
_MAIN_AUX = _os.environ.get("CTI_SIBLINGS_MAIN_AUX", "0") == "1"
_no_nested_fork = [False]       # set while run_concurrently's functions run: their models fork no auxiliary stream of their own


def run_concurrently(*fns):
    """Independent forwards (e.g. the BAN and the CTI teacher of BASELINE configs[3]: two models, one batch, nothing shared but the inputs) on
    sibling streams: fns[0] on the caller's stream, every other one on its own side stream forked from it here and joined before returning, so a
    chain of small dependent launches in one model fills the compute units another model's chain leaves idle.  Results in call order.  Works eagerly and
    under hipGraph capture (fork / join become graph edges).  Inference only: under autograd the backward would run on the side streams' graph.
    The functions run WITHOUT the library's auxiliary stream: with two chains in flight a third adds contention, not overlap (configs[3]: 127.8 k
    samples/s without, 121.5-125.0 k with it on either model, profiles/r04_c4_concurrent_models.txt) -- and ending a capture in which a sibling
    stream forks again takes this ROCm runtime down (hipStreamEndCapture segfault, measured in round 4)."""
    if len(fns) < 2 or torch.is_grad_enabled():
        return tuple(f() for f in fns)
    cur = torch.cuda.current_stream()
    from . import fc as _fc
    _fc.refresh_stale_scales(cur.device)             # the batched weight-norm refresh belongs before the fork (it serves layers of every model)
    key = (cur.device.index, len(fns) - 1)
    if key not in _sib:
        _sib[key] = [torch.cuda.Stream(device=cur.device) for _ in fns[1:]]
    outs = [None] * len(fns)
    capturing = torch.cuda.is_current_stream_capturing()

    def no_aux(f):
        _no_nested_fork[0] = True
        try:
            return f()
        finally:
            _no_nested_fork[0] = False
    for i, s in enumerate(_sib[key]):
        s.wait_stream(cur)
        with torch.cuda.stream(s):
            outs[i + 1] = no_aux(fns[i + 1])
    # (round 6, re-measured: CTI_SIBLINGS_MAIN_AUX=1 lets fns[0] -- on the caller's stream, which may fork as any single model's capture does -- keep its auxiliary
    # streams: configs[3] 1 330 -> 1 448 us.  A third chain still adds contention, not overlap; default off)
    outs[0] = fns[0]() if _MAIN_AUX else no_aux(fns[0])
    for i, s in enumerate(_sib[key]):
        cur.wait_stream(s)
        if not capturing:                        # (a capture's private pool keeps its blocks for the graph's lifetime)
            for t in _tensors_of(outs[i + 1]):
                t.record_stream(cur)             # allocated on the side stream, consumed on the caller's
    return tuple(outs)


def _tensors_of(o):
    if isinstance(o, torch.Tensor):
        return [o]
    if isinstance(o, (tuple, list)):
        return [t for x in o for t in _tensors_of(x)]
    return []


# ---- optional per-kernel timing with HIP events on the launch stream (bench.py's roofline leg) -----------------
_prof = None


def profile_start():
    """Start recording a (start, end) event pair around every launch family, on the stream it is launched on."""
    global _prof
    _prof = {}


def profile_stop():
    """Stop recording; returns {name: [ms, ...]} (synchronises)."""
    global _prof
    rec, _prof = _prof, None
    torch.cuda.synchronize()
    return {k: [pr.ms() if isinstance(pr, _LibEventPair) else pr[0].elapsed_time(pr[1]) for pr in v] for k, v in (rec or {}).items()}


class _LibEventPair:
    """Two hipEvent_t handles owned through the C ABI (cti_event_*)."""

    def __init__(self, a, b):
        self.a, self.b = a, b

    def ms(self):
        import ctypes as C
        out = C.c_float(0)
        lib = L.lib()
        L.check(lib.cti_event_elapsed_ms(self.a, self.b, C.byref(out)), "cti_event_elapsed_ms")
        lib.cti_event_destroy(self.a); lib.cti_event_destroy(self.b)
        return float(out.value)


class _timed:
    def __init__(self, name):
        self.name = name

    def __enter__(self):
        if _prof is not None:
            self.a = torch.cuda.Event(enable_timing=True)
            self.b = torch.cuda.Event(enable_timing=True)
            self.a.record(torch.cuda.current_stream())

    def __exit__(self, *exc):
        if _prof is not None:
            self.b.record(torch.cuda.current_stream())
            _prof.setdefault(self.name, []).append((self.a, self.b))
        return False


def _req(t, name, dtype=torch.float32):
    if not isinstance(t, torch.Tensor):
        raise TypeError("%s must be a tensor" % name)
    if not t.is_cuda:
        raise L.CtiError("%s is on %s: the CTI modules run on an MI355X through libcti_hip.so only (no CPU path)" % (name, t.device))
    if t.dtype != dtype:
        raise TypeError("%s must be %s, got %s" % (name, dtype, t.dtype))
    return t


def _rows2d(x):
    """(…, d) -> contiguous-row 2-D view (rows, d) with a row stride; copies only when the layout demands it."""
    d = x.shape[-1]
    if x.dim() == 2 and x.stride(1) == 1 and x.stride(0) >= d:
        return x, x.stride(0)
    if not x.is_contiguous():
        x = x.contiguous()
    return x.view(-1, d), d


def _ptr(t):
    return 0 if t is None else t.data_ptr()


def wn_scale(weight_v, weight_g):
    """scale[i] = g[i] / ||V_i||_F.  weight_v: (n_mats, ...) contiguous or a single matrix with scalar g."""
    _req(weight_v, "weight_v"); _req(weight_g, "weight_g")
    n = weight_g.numel()
    wv = weight_v.contiguous()
    g = weight_g.contiguous().view(-1)
    out = torch.empty(n, device=wv.device, dtype=torch.float32)
    lib = L.lib()
    wsb = lib.cti_wn_scale_workspace_bytes(n, wv.numel() // n)
    ws = torch.empty(wsb, device=wv.device, dtype=torch.uint8) if wsb else None
    L.check(lib.cti_wn_scale(wv.data_ptr(), g.data_ptr(), out.data_ptr(), n, wv.numel() // n, _ptr(ws), wsb, _stream()), "cti_wn_scale")
    return out


def wn_scale_many(pairs):
    """[(weight_v, weight_g), ...] (single matrices, scalar g, one device) -> a tensor of len(pairs) scales from ONE pair of launches per 48
    layers (cti_wn_scale_many)."""
    import ctypes as C
    n = len(pairs)
    vs = [_req(v, "weight_v").contiguous() for v, _ in pairs]
    gs = [_req(g, "weight_g").contiguous().view(-1) for _, g in pairs]
    dev = vs[0].device
    out = torch.empty(n, device=dev, dtype=torch.float32)
    pv = (C.c_void_p * n)(*[v.data_ptr() for v in vs])
    pg = (C.c_void_p * n)(*[g.data_ptr() for g in gs])
    ps = (C.c_void_p * n)(*[out.data_ptr() + 4 * i for i in range(n)])
    el = (C.c_int64 * n)(*[v.numel() for v in vs])
    lib = L.lib()
    wsb = lib.cti_wn_scale_many_workspace_bytes(el, n)
    ws = torch.empty(max(wsb, 16), device=dev, dtype=torch.uint8)
    L.check(lib.cti_wn_scale_many(pv, pg, ps, el, n, ws.data_ptr(), wsb, _stream()), "cti_wn_scale_many")
    return out


def wn_linear(x, weight_v, scale, scale_div, bias, relu, prec=None, w_planes=None):
    """act(scale[n // scale_div] * x @ weight_v.T + bias) through the MFMA GEMM.  w_planes: split_operand(weight_v) kept by the caller."""
    _req(x, "x"); _req(weight_v, "weight_v")
    out_dim, in_dim = weight_v.shape
    if x.shape[-1] != in_dim:
        raise ValueError("x has %d features, the layer expects %d" % (x.shape[-1], in_dim))
    x2, ldx = _rows2d(x)
    w = weight_v if (weight_v.stride(1) == 1) else weight_v.contiguous()
    rows = x2.shape[0]
    y = torch.empty(x.shape[:-1] + (out_dim,), device=x.device, dtype=torch.float32)
    if rows == 0:
        return y
    pr = _prec(prec)
    lib = L.lib()
    if out_dim <= 8 and scale_div >= out_dim and not torch.is_grad_enabled():
        # a handful of outputs (the MC models' answer head: 2): one exact-fp32 kernel instead of a split-K GEMM launch pair
        rc = lib.cti_linear_small_n(x2.data_ptr(), ldx, w.data_ptr(), w.stride(0), _ptr(scale), _ptr(bias), y.data_ptr(), out_dim, rows, in_dim, out_dim, int(bool(relu)),
                                    _stream())
        if rc != L.E_UNSUPPORTED:
            L.check(rc, "cti_linear_small_n")
            return y
    if w_planes is not None and pr != L.PREC_F32:
        y2 = y.view(-1, out_dim)
        gemm_nt(x2, w, M=rows, N=out_dim, out=y2, c_strides=(out_dim, 1), scale=scale, scale_div=scale_div, bias=bias, relu=relu, prec=prec,
                B_planes=w_planes)
        return y
    wsb = lib.cti_wn_linear_workspace_bytes(rows, in_dim, out_dim, pr)
    ws = torch.empty(wsb, device=x.device, dtype=torch.uint8) if wsb else None
    with _timed("wn_linear_%dx%dx%d" % (rows, in_dim, out_dim)):
      L.check(lib.cti_wn_linear_fwd(x2.data_ptr(), ldx, w.data_ptr(), w.stride(0), _ptr(scale), int(scale_div), _ptr(bias),
                                  y.data_ptr(), out_dim, rows, in_dim, out_dim, L.ACT_RELU if relu else L.ACT_NONE, pr,
                                  _ptr(ws), wsb, _stream()), "cti_wn_linear_fwd")
    return y


def zero_row_mask(v):
    """(B, V, d) -> uint8 (B, V): 1 where the row is all zeros (reference `0 == v.abs().sum(2)`)."""
    _req(v, "v", v.dtype if isinstance(v, torch.Tensor) and v.dtype == torch.bfloat16 else torch.float32)
    x2, ld = _rows2d(v)
    mask = torch.empty(v.shape[:-1], device=v.device, dtype=torch.uint8)
    if mask.numel() == 0:
        return mask
    if v.dtype == torch.bfloat16:                                    # (round 5: bf16 inputs of the plain-bf16 mode; same bit-exact rule: every element +-0)
        L.check(L.lib().cti_zero_row_mask_bf16(x2.data_ptr(), ld, mask.data_ptr(), x2.shape[0], v.shape[-1], _stream()), "cti_zero_row_mask_bf16")
        return mask
    L.check(L.lib().cti_zero_row_mask(x2.data_ptr(), ld, mask.data_ptr(), x2.shape[0], v.shape[-1], _stream()), "cti_zero_row_mask")
    return mask


def teff_scramble(T, inverse=False):
    """T: (R, I, J, K, G) contiguous -> T_eff (same shape); inverse=True maps a T_eff-shaped tensor back."""
    _req(T, "T")
    T = T.contiguous()
    R, I, J, K, G = T.shape
    out = torch.empty_like(T)
    L.check(L.lib().cti_teff_scramble(T.data_ptr(), out.data_ptr(), R, I, J, K, G, 1 if inverse else 0, _stream()), "cti_teff_scramble")
    return out


def paralind_mbuild(Vr, Qr, Teff):
    """Vr (B,V,R*I), Qr (B,Q,R*J), Teff (R,I,J,K,G) -> M (B,V,Q,G,R*K)."""
    _req(Vr, "Vr"); _req(Qr, "Qr"); _req(Teff, "Teff")
    R, I, J, K, G = Teff.shape
    B, V, _ = Vr.shape
    Q = Qr.shape[1]
    assert Vr.shape[2] == R * I and Qr.shape[2] == R * J and Qr.shape[0] == B
    Vr, Qr, Teff = Vr.contiguous(), Qr.contiguous(), Teff.contiguous()
    M = torch.empty((B, V, Q, G, R * K), device=Vr.device, dtype=torch.float32)
    if M.numel() == 0:
        return M
    with _timed("paralind_mbuild"):
      L.check(L.lib().cti_paralind_mbuild_fwd(Vr.data_ptr(), Qr.data_ptr(), Teff.data_ptr(), M.data_ptr(), B, V, Q, R, I, J, K, G,
                                            _stream()), "cti_paralind_mbuild_fwd")
    return M


def paralind_mbuild_planes(Vr, Qr, Teff, use_mfma=True):
    """The same M as bf16 hi/lo operand planes of the mode-3 GEMM: returns (Mh, Ml) int16 tensors (K/16, rows_alloc, 16) with
    rows_alloc = B*V*Q*G + 256, element (row, k) at [k >> 4, row, k & 15].  This is what the fused TCNet.forward feeds its last GEMM."""
    _req(Vr, "Vr"); _req(Qr, "Qr"); _req(Teff, "Teff")
    R, I, J, K, G = Teff.shape
    if not (I == J == K):
        raise ValueError("the plane-writing M build needs a cubic core, got %s" % (tuple(Teff.shape),))
    B, V, _ = Vr.shape
    Q = Qr.shape[1]
    if (R * K) % 32:
        raise ValueError("R*hr = %d must be a multiple of 32 (the planes' K padding)" % (R * K))
    Vr, Qr, Teff = Vr.contiguous(), Qr.contiguous(), Teff.contiguous()
    rows_alloc = B * V * Q * G + 256
    Mh = torch.empty((R * K // 16, rows_alloc, 16), device=Vr.device, dtype=torch.int16)
    Ml = torch.empty_like(Mh)
    Mh[:, rows_alloc - 256:, :].zero_()                                    # only the tile over-read slack needs defined contents
    Ml[:, rows_alloc - 256:, :].zero_()
    Tt = transpose(Teff, I, J * K * G, batch=R, s_src=I * J * K * G, ld_src=J * K * G, s_dst=I * J * K * G, ld_dst=I).view(R, J * K * G, I) if use_mfma else None
    L.check(L.lib().cti_paralind_mbuild_planes_fwd(Vr.data_ptr(), Qr.data_ptr(), Teff.data_ptr(), _ptr(Tt), Mh.data_ptr(), Ml.data_ptr(), B, V, Q, R, I, G,
                                                   rows_alloc, _stream()), "cti_paralind_mbuild_planes_fwd")
    return Mh, Ml


def paralind_mbuild_f16f6(Vr, Qr, Teff):
    """The same M as the f16f6 operand block of the mode-3 product (rows (b,v,q,g) in batches of V*Q*G), encoded inside the M build."""
    _req(Vr, "Vr"); _req(Qr, "Qr"); _req(Teff, "Teff")
    R, I, J, K, G = Teff.shape
    B, V, _ = Vr.shape
    Q = Qr.shape[1]
    Vr, Qr, Teff = Vr.contiguous(), Qr.contiguous(), Teff.contiguous()
    Tt = transpose(Teff, I, J * K * G, batch=R, s_src=I * J * K * G, ld_src=J * K * G, s_dst=I * J * K * G, ld_dst=I).view(R, J * K * G, I)
    lib = L.lib()
    nb = lib.cti_f16f6_planes_bytes(B * V * Q * G, R * K, V * Q * G)
    block = torch.zeros(nb, device=Vr.device, dtype=torch.uint8)
    L.check(lib.cti_paralind_mbuild_f16f6_fwd(Vr.data_ptr(), Qr.data_ptr(), Tt.data_ptr(), block.data_ptr(), nb, B, V, Q, R, I, G, _stream()),
            "cti_paralind_mbuild_f16f6_fwd")
    return block


def paralind_core(M, Ar, prec=None):
    """M (B,V,Q,G,K), Ar (B,A,K) -> out (B,V,Q,A,G) contiguous."""
    _req(M, "M"); _req(Ar, "Ar")
    B, V, Q, G, K = M.shape
    A = Ar.shape[1]
    assert Ar.shape[0] == B and Ar.shape[2] == K
    M, Ar = M.contiguous(), Ar.contiguous()
    out = torch.empty((B, V, Q, A, G), device=M.device, dtype=torch.float32)
    if out.numel() == 0:
        return out
    pr = _prec(prec)
    lib = L.lib()
    wsb = lib.cti_paralind_core_workspace_bytes(B, V * Q, A, G, K, pr)
    ws = torch.empty(wsb, device=M.device, dtype=torch.uint8) if wsb else None
    with _timed("paralind_core"):
      L.check(lib.cti_paralind_core_fwd(M.data_ptr(), Ar.data_ptr(), out.data_ptr(), B, V * Q, A, G, K, pr, _ptr(ws), wsb, _stream()),
            "cti_paralind_core_fwd")
    return out


def paralind_core_planes(Mh, Ml, Ar, B, V, Q, G, prec=None):
    """Mode 3 + rank sum with M given as operand planes (paralind_mbuild_planes): out (B,V,Q,A,G)."""
    _req(Ar, "Ar")
    Ar = Ar.contiguous()
    A, K = Ar.shape[1], Ar.shape[2]
    out = torch.empty((B, V, Q, A, G), device=Ar.device, dtype=torch.float32)
    pr = _prec(prec)
    lib = L.lib()
    wsb = lib.cti_paralind_core_planes_workspace_bytes(B, A, K, pr)
    ws = torch.empty(max(wsb, 16), device=Ar.device, dtype=torch.uint8)
    L.check(lib.cti_paralind_core_planes_fwd(Mh.data_ptr(), Ml.data_ptr(), Mh.shape[1], Ar.data_ptr(), out.data_ptr(), B, V * Q, A, G, K, pr,
                                             ws.data_ptr(), wsb, _stream()), "cti_paralind_core_planes_fwd")
    return out


def paralind_core_bwd_planes(dout, Mh, Ml, Ar, G):
    """dM (B,V,Q,G,K) and dAr (B,A,K) from dout (B,V,Q,A,G) with M held as operand planes."""
    B, V, Q, A, _ = dout.shape
    K = Ar.shape[2]
    dout, Ar = _req(dout, "dout").contiguous(), Ar.contiguous()
    dM = torch.empty((B, V, Q, G, K), device=dout.device, dtype=torch.float32)
    dAr = torch.empty_like(Ar)
    L.check(L.lib().cti_paralind_core_bwd_planes(dout.data_ptr(), Mh.data_ptr(), Ml.data_ptr(), Mh.shape[1], Ar.data_ptr(), dM.data_ptr(), dAr.data_ptr(),
                                                 B, V, Q, A, G, K, _stream()), "cti_paralind_core_bwd_planes")
    return dM, dAr


def _weight_arrays(tucker, rank):
    import ctypes as C
    keep = []

    def arr(ts):
        ts = [_req(t, "weight").contiguous() for t in ts]
        keep.extend(ts)
        return (C.c_void_p * 3)(*[t.data_ptr() for t in ts])
    return (arr([t[0] for t in tucker]), arr([t[1] for t in tucker]), arr([t[2] for t in tucker]),
            arr([t[0] for t in rank]), arr([t[1] for t in rank]), arr([t[2] for t in rank]), keep)


def tcnet_prepare(tucker, rank, T_g, prec=None):
    """The batch-independent part of tcnet_forward (weight-norm scales, T_eff, the weights' operand planes) as one device block: compute it
    once per parameter update and pass it as `prepared=`.  Returns (block, precision code it was built for)."""
    _req(T_g, "T_g")
    h = tucker[0][0].shape[0]
    R = rank[0][1].numel()
    G = T_g.shape[5]
    vd, qd, ad = (tucker[s][0].shape[1] for s in range(3))
    pr = _prec(prec, fused=True)
    if pr == L.PREC_F16F6 and h % 32:
        pr = L.PREC_BF16X3                                       # the f16f6 planes need h % 32 == 0
    lib = L.lib()
    nb = lib.cti_tcnet_prepared_bytes(vd, qd, ad, h, R, G, pr)
    block = torch.empty(nb, device=T_g.device, dtype=torch.uint8)
    twv, tg, tb, rwv, rg, rb, keep = _weight_arrays(tucker, rank)
    L.check(lib.cti_tcnet_prepare(twv, tg, rwv, rg, T_g.contiguous().data_ptr(), vd, qd, ad, h, R, G, pr, block.data_ptr(), nb, _stream()),
            "cti_tcnet_prepare")
    return block, pr


_debug_ws_fill = None      # tests only: byte value the fused forward's workspace is filled with before the call


def tcnet_forward(v, q, a, tucker, rank, T_g, relu=True, want_mask=False, prec=None, prepared=None, want_sm_partials=False, _tri=False, _v_tucked=None, _v_rep=1):
    """Whole TCNet.forward in one C-ABI call.  tucker / rank: 3-lists (v, q, a order) of (weight_v, weight_g, bias);
    the rank entries are PACKED: weight_v (h, h), weight_g (R,), bias (h,).  prepared: the (block, precision) pair of tcnet_prepare for
    these weights (optional).  Returns out (B,V,Q,A,G) [, mask (B,V)] [, partials]: with want_sm_partials (needs want_mask) the third
    value is the Tri softmax's partial pass left by the mode-3 GEMM (masked_softmax_tri_from_partials_), or None where the library has no
    fused form for this precision / glimpse.  _tri (triattention_forward): cti_triattention_forward instead -- returns (p, logits)."""
    # (round 5) bf16 `v` + bf16 hoisted projection: TriAttention's few-answer path takes them as they are (cti_triattention_forward_vt16: v is read for the
    # zero-row mask only); every other path widens them first
    v16 = isinstance(v, torch.Tensor) and v.dtype == torch.bfloat16
    if v16 and not (_tri and _v_tucked is not None and _v_tucked.dtype == torch.bfloat16):
        v, v16 = widen_bf16(v), False
    if not v16 and _v_tucked is not None and _v_tucked.dtype == torch.bfloat16:
        _v_tucked = widen_bf16(_v_tucked)
    _req(v, "v", torch.bfloat16 if v16 else torch.float32)
    for t, n in ((q, "q"), (a, "a"), (T_g, "T_g")):
        _req(t, n)
    v, q, a = v.contiguous(), q.contiguous(), a.contiguous()
    B, V, vd = v.shape
    Q, qd = q.shape[1], q.shape[2]
    A, ad = a.shape[1], a.shape[2]
    if q.shape[0] != B or a.shape[0] != B:
        raise ValueError("v, q, a must share the batch dimension")
    h = tucker[0][0].shape[0]
    R = rank[0][1].numel()
    G = T_g.shape[5]
    if T_g.shape[6] != 1:
        raise RuntimeError("TCNet.forward: h_out must be 1 (src/Tensor.py:6 cannot view the core otherwise)")
    for s, dim in enumerate((vd, qd, ad)):
        if tuple(tucker[s][0].shape) != (h, dim) or tuple(rank[s][0].shape) != (h, h):
            raise ValueError("weight shapes do not match the inputs")
    twv, tg, tb, rwv, rg, rb, keep = _weight_arrays(tucker, rank)
    Tg = T_g.contiguous()
    out = torch.empty((B, V, Q, A, G), device=v.device, dtype=torch.float32)
    if _tri:
        want_mask, want_sm_partials = True, False
    mask = torch.empty((B, V), device=v.device, dtype=torch.uint8) if want_mask else None
    if want_sm_partials and not want_mask:
        raise ValueError("want_sm_partials needs want_mask")
    if out.numel() == 0:                                   # empty batch (or a zero-length axis): nothing to launch
        if _tri:
            return torch.empty_like(out), out
        return ((out, mask, None) if want_sm_partials else (out, mask)) if want_mask else out
    pr = _prec(prec, fused=True)
    if pr == L.PREC_F16F6 and h % 32:
        pr = L.PREC_BF16X3
    prep_ptr = 0
    if prepared is not None:
        if prepared[1] != pr:
            raise ValueError("the prepared block was built for another precision mode")
        prep_ptr = prepared[0].data_ptr()
    lib = L.lib()
    guarded = pr == L.PREC_F16F6 and lib.cti_tcnet_forward_guard_bytes(B, V, Q, A, vd, qd, ad, h, R, G, pr) > 0
    wait_guard = guarded and _range_check == "sync" and not torch.cuda.is_current_stream_capturing()
    own = _owner_state(v.device, T_g) if wait_guard else None
    if wait_guard and own["skip"] > 0:
        # THIS network keeps leaving the format's domain (two trips in a row): go straight to bf16x3 for a while instead of paying for both forms
        own["skip"] -= 1
        _range_log["skip"] = own["skip"]
        return tcnet_forward(v, q, a, tucker, rank, T_g, relu, want_mask, own["skip_mode"], None, want_sm_partials, _tri, _v_tucked, _v_rep)
    wsb = (lib.cti_triattention_workspace_bytes if _tri else lib.cti_tcnet_forward_workspace_bytes)(B, V, Q, A, vd, qd, ad, h, R, G, pr)
    ws = torch.empty(wsb, device=v.device, dtype=torch.uint8)
    if _debug_ws_fill is not None:              # tests: the library must not read workspace bytes it has not written (0xFF reads as saturated scale bytes)
        ws.fill_(_debug_ws_fill)
    ev0 = ev1 = None
    if _prof is not None:                       # hipEvents around the mode-3 GEMM, recorded by the library on the launch stream
        ev0, ev1 = lib.cti_event_create(), lib.cti_event_create()
        _prof.setdefault("paralind_core", []).append(_LibEventPair(ev0, ev1))
    elif wait_guard:
        ev0, ev1 = _guard_resources(v.device)[:2]
    part = p_att = None
    if want_sm_partials:
        pb = lib.cti_tcnet_softmax_partials_bytes(B, V, Q, A, h, G, pr)
        if pb and out.data_ptr() % 16 == 0 and (V * Q * A) % 2 == 0:
            part = torch.empty(pb, device=v.device, dtype=torch.uint8)
    act = L.ACT_RELU if relu else L.ACT_NONE
    if guarded:
        # a call whose verdict the host reads is re-run on ANY bit (the NaN fill is belt and braces); one nobody waits for keeps its f16f6 result on the
        # cancellation bits (set_range_check)
        lib.cti_set_tuning(L.TUNE_GUARD_POISON_BITS, 31 if wait_guard else _poison_bits_nohost)
    with _timed("triattention_forward" if _tri else "tcnet_forward"):
        if _tri:
            p_att = torch.empty_like(out)
            vt_ptr, vt_ld, vt_rep = 0, 0, 1
            if _v_tucked is not None and lib.cti_triattention_hoist_ok(B, V, Q, A, h, R, G, pr) and B % int(_v_rep) == 0:
                # relu(v_tucker(v)) from the caller's batched projection (one block per image when _v_rep > 1): rows (B / rep * V, >= h), 16-B aligned
                vt2 = _v_tucked.reshape(-1, _v_tucked.shape[-1])
                if (vt2.stride(1) == 1 and vt2.shape[0] == (B // int(_v_rep)) * V and vt2.shape[1] >= h and vt2.stride(0) % (8 if v16 else 4) == 0
                        and vt2.data_ptr() % 16 == 0 and vt2.dtype == (torch.bfloat16 if v16 else torch.float32) and (not v16 or (h % 8 == 0 and vd % 2 == 0))):
                    vt_ptr, vt_ld, vt_rep = vt2.data_ptr(), vt2.stride(0), int(_v_rep)
            if v16 and vt_ptr == 0:                                   # the bf16 form needs the hoisted projection: widen and take the fp32 call
                v, v16 = widen_bf16(v), False
                if _v_tucked is not None:
                    return tcnet_forward(v, q, a, tucker, rank, T_g, relu, want_mask, prec, prepared, want_sm_partials, _tri, widen_bf16(_v_tucked), _v_rep)
            if v16:
                L.check(lib.cti_triattention_forward_vt16(v.data_ptr(), q.data_ptr(), a.data_ptr(), twv, tg, tb, rwv, rg, rb, Tg.data_ptr(), out.data_ptr(),
                                                          p_att.data_ptr(), mask.data_ptr(), B, V, Q, A, vd, qd, ad, h, R, G, act, pr, prep_ptr, ws.data_ptr(), wsb,
                                                          ev0, ev1, _aux_stream(v.device), _stream(), vt_ptr, vt_ld, vt_rep), "cti_triattention_forward_vt16")
            else:
              L.check(lib.cti_triattention_forward(v.data_ptr(), q.data_ptr(), a.data_ptr(), twv, tg, tb, rwv, rg, rb, Tg.data_ptr(), out.data_ptr(),
                                                 p_att.data_ptr(), mask.data_ptr(), B, V, Q, A, vd, qd, ad, h, R, G, act, pr, prep_ptr, ws.data_ptr(), wsb,
                                                 ev0, ev1, _aux_stream(v.device), _stream(), vt_ptr, vt_ld, vt_rep), "cti_triattention_forward")
        elif part is not None:
            L.check(lib.cti_tcnet_forward_sm(v.data_ptr(), q.data_ptr(), a.data_ptr(), twv, tg, tb, rwv, rg, rb, Tg.data_ptr(), out.data_ptr(),
                                             _ptr(mask), B, V, Q, A, vd, qd, ad, h, R, G, act, pr,
                                             prep_ptr, ws.data_ptr(), wsb, ev0, ev1, _aux_stream(v.device), _stream(), part.data_ptr(), pb),
                    "cti_tcnet_forward_sm")
        else:
            L.check(lib.cti_tcnet_forward(v.data_ptr(), q.data_ptr(), a.data_ptr(), twv, tg, tb, rwv, rg, rb, Tg.data_ptr(), out.data_ptr(),
                                          _ptr(mask), B, V, Q, A, vd, qd, ad, h, R, G, act, pr,
                                          prep_ptr, ws.data_ptr(), wsb, ev0, ev1, _aux_stream(v.device), _stream()), "cti_tcnet_forward")
    if guarded:
        _range_log["calls"] += 1
        if not wait_guard:
            _keep_guard_block(ws)          # nobody waits for this call's verdict: leave it where f16f6_device_status() / GraphedForward find it
    if wait_guard:
        import ctypes as _C
        status = _C.c_uint32(0)
        ev_verdict, aux = ev0, _aux_stream(v.device)
        if aux is not None:
            # with an auxiliary stream the guard's kernels run there, BESIDE the mode-3 product (round 6), and are the last work the call leaves on it:
            # an event recorded on it now marks the verdict (include/cti_hip.h, "Range guard")
            ev_verdict = _guard_resources(v.device)[3]
            L.check(lib.cti_event_record(ev_verdict, aux), "cti_event_record")
        L.check(lib.cti_guard_read(ws.data_ptr(), ev_verdict, _guard_resources(v.device)[2].cuda_stream, _C.byref(status)), "cti_guard_read")
        _range_log["last_status"] = int(status.value)
        if _range_debug:
            ratio = _C.c_float(0.0)
            L.check(lib.cti_guard_read_ratio(ws.data_ptr(), _guard_resources(v.device)[2].cuda_stream, _C.byref(ratio)), "cti_guard_read_ratio")
            _range_log["last_ratio"] = float(ratio.value)
        if status.value:
            # an operand left the f16f6 format's domain (the output of this launch has been NaN-filled on the device): the reference's
            # full-range fp32 semantics come from the bf16x3 kernels
            _range_log["trips"] += 1
            own["consecutive"] += 1
            _range_log["consecutive"] = own["consecutive"]
            if own["consecutive"] >= 2:
                own["skip"] = _range_log["skip"] = 64
                own["skip_mode"] = _range_log["skip_mode"] = "fp32" if status.value & 16 else "bf16x3"
            if _range_log["trips"] == 1:
                import warnings
                warnings.warn("cti: f16f6 range guard tripped (status %d: %s) -- this call was re-run in the %s mode; see ops.f16f6_range_status()"
                              % (status.value, ", ".join(n for b, n in ((1, "saturation / non-finite"), (2, "underflow"), (4, "non-finite V^/Q^/T"),
                                                                        (8, "heavy cancellation"), (16, "very heavy cancellation")) if status.value & b),
                                 "fp32" if status.value & 16 else "bf16x3"))
            # (heavy cancellation: even the 3-term split's 2^-19 constant may exceed the tolerance -- the exact-fp32 kernels are the reference's arithmetic)
            return tcnet_forward(v, q, a, tucker, rank, T_g, relu, want_mask, "fp32" if status.value & 16 else "bf16x3", None, want_sm_partials, _tri, _v_tucked, _v_rep)
        own["consecutive"] = _range_log["consecutive"] = 0
    if _tri:
        return p_att, out
    if want_sm_partials:
        return out, mask, part
    return (out, mask) if want_mask else out


def triattention_hoist_ok(B, V, Q, A, h, R, G, prec=None):
    """True when cti_triattention_forward takes a hoisted v projection at this shape and precision (its fused few-answer path)."""
    pr = _prec(prec, fused=True)
    if pr == L.PREC_F16F6 and h % 32:
        pr = L.PREC_BF16X3
    return bool(L.lib().cti_triattention_hoist_ok(int(B), int(V), int(Q), int(A), int(h), int(R), int(G), pr))


def triattention_forward(v, q, a, tucker, rank, T_g, relu=True, prec=None, prepared=None, v_tucked=None, v_rep=1):
    """TriAttention.forward (reference src/attention.py:49-59) in ONE C-ABI call (cti_triattention_forward): returns (p, logits), both
    (B,V,Q,A,G), logits with -inf on the all-zero rows of v.  Arguments as tcnet_forward.  v_tucked (optional): relu(v_tucker(v)) from the
    caller's batched projection, (B / v_rep, V, >= h) fp32 -- used where the library takes it (few answer tokens), ignored elsewhere."""
    return tcnet_forward(v, q, a, tucker, rank, T_g, relu, True, prec, prepared, False, True, v_tucked, v_rep)


def masked_softmax_tri_from_partials_(logits, mask, partials):
    """masked_softmax_tri_ when tcnet_forward(want_sm_partials=True) already ran the partial pass: one read of the logits instead of two."""
    _req(logits, "logits"); _req(mask, "mask", torch.uint8)
    assert logits.is_contiguous() and mask.is_contiguous()
    B, V, Q, A, G = logits.shape
    p = torch.empty_like(logits)
    ws = torch.empty(B * G * 2, device=logits.device, dtype=torch.float32)
    with _timed("masked_softmax_tri"):
        L.check(L.lib().cti_masked_softmax_tri_from_partials_fwd(logits.data_ptr(), mask.data_ptr(), partials.data_ptr(), partials.numel(), p.data_ptr(),
                                                                 B, V, Q * A, G, ws.data_ptr(), ws.numel() * 4, _stream()),
                "cti_masked_softmax_tri_from_partials_fwd")
    return p


def masked_softmax_tri_(logits, mask):
    """In place -inf fill of `logits` (B,V,Q,A,G contiguous) on masked rows; returns p (same shape)."""
    _req(logits, "logits"); _req(mask, "mask", torch.uint8)
    assert logits.is_contiguous() and mask.is_contiguous()
    B, V, Q, A, G = logits.shape
    p = torch.empty_like(logits)
    if p.numel() == 0:
        return p
    lib = L.lib()
    wsb = lib.cti_softmax_tri_workspace_bytes(B, V, Q * A, G)
    ws = torch.empty(wsb, device=logits.device, dtype=torch.uint8)
    with _timed("masked_softmax_tri"):
      L.check(lib.cti_masked_softmax_tri_fwd(logits.data_ptr(), mask.data_ptr(), p.data_ptr(), B, V, Q * A, G, ws.data_ptr(), wsb,
                                           _stream()), "cti_masked_softmax_tri_fwd")
    return p


def masked_softmax_bi_(logits, mask):
    """logits (B,G,V,Q) contiguous, mask (B,V) uint8 or None; in place -inf fill; returns p."""
    _req(logits, "logits")
    assert logits.is_contiguous()
    if mask is not None:
        _req(mask, "mask", torch.uint8)
    B, G, V, Q = logits.shape
    p = torch.empty_like(logits)
    if p.numel() == 0:
        return p
    L.check(L.lib().cti_masked_softmax_bi_fwd(logits.data_ptr(), _ptr(mask), p.data_ptr(), B, G, V, Q, _stream()),
            "cti_masked_softmax_bi_fwd")
    return p


def tri_pool(vt, qt, at, w, v_rep=1):
    """out[b,d] = sum_vqa vt[b,v,d] w[b,v,q,a] qt[b,q,d] at[b,a,d]; w may be any strided (B,V,Q,A) view.  v_rep > 1: vt is (B / v_rep, V, D),
    one block per image shared by v_rep consecutive batch rows (the kernel that takes it reads it in place; otherwise it is expanded here)."""
    if isinstance(vt, torch.Tensor) and vt.dtype == torch.bfloat16:
        vt = widen_bf16(vt)                                            # (the unshifted pools read fp32 rows)
    for t, n in ((vt, "vt"), (qt, "qt"), (at, "at"), (w, "w")):
        _req(t, n)
    v_rep = int(v_rep)
    B, (V, D) = qt.shape[0], vt.shape[1:]
    if vt.shape[0] * v_rep != B:
        raise ValueError("vt has %d blocks for a batch of %d with v_rep=%d" % (vt.shape[0], B, v_rep))
    Q, A = qt.shape[1], at.shape[1]
    if tuple(w.shape) != (B, V, Q, A):
        raise ValueError("w must be (B,V,Q,A) = %s, got %s" % ((B, V, Q, A), tuple(w.shape)))
    vt, qt, at = vt.contiguous(), qt.contiguous(), at.contiguous()
    out = torch.empty((B, D), device=vt.device, dtype=torch.float32)
    if B == 0:
        return out
    if V * Q * A == 0:
        return out.zero_()
    sb, sv, sq, sa = w.stride()
    lib = L.lib()
    if get_precision() != "fp32" and _os.environ.get("CTI_NO_TRI_POOL_MFMA", "0") != "1":      # fp32-grade MFMA form; exact-fp32 mode keeps the VALU kernels
        rc = lib.cti_tri_pool_mfma_fwd(vt.data_ptr(), qt.data_ptr(), at.data_ptr(), w.data_ptr(), sb, sv, sq, sa, out.data_ptr(), B, V, Q, A, D,
                                       v_rep, _stream())
        if rc != L.E_UNSUPPORTED:
            L.check(rc, "cti_tri_pool_mfma_fwd")
            return out
    if v_rep > 1:
        vt = vt.repeat_interleave(v_rep, 0)
    L.check(lib.cti_tri_pool_fwd(vt.data_ptr(), qt.data_ptr(), at.data_ptr(), w.data_ptr(), sb, sv, sq, sa, out.data_ptr(),
                                 B, V, Q, A, D, _stream()), "cti_tri_pool_fwd")
    return out


def bi_pool(vt, qt, w, k=1):
    """out[b,n] = sum_{t<k} sum_vq vt[b,v,nk+t] w[b,v,q] qt[b,q,nk+t]; w=None means all ones."""
    if isinstance(vt, torch.Tensor) and vt.dtype == torch.bfloat16:
        vt = widen_bf16(vt)                                            # (the unshifted pools read fp32 rows: as tri_pool)
    if isinstance(qt, torch.Tensor) and qt.dtype == torch.bfloat16:
        qt = widen_bf16(qt)
    _req(vt, "vt"); _req(qt, "qt")
    B, V, D = vt.shape
    Q = qt.shape[1]
    vt, qt = vt.contiguous(), qt.contiguous()
    if w is not None:
        _req(w, "w")
        if tuple(w.shape) != (B, V, Q):
            raise ValueError("w must be (B,V,Q) = %s, got %s" % ((B, V, Q), tuple(w.shape)))
        sb, sv, sq = w.stride()
    else:
        sb = sv = sq = 0
    out = torch.empty((B, D // k), device=vt.device, dtype=torch.float32)
    if B == 0:
        return out
    if V * Q == 0:
        return out.zero_()
    L.check(L.lib().cti_bi_pool_fwd(vt.data_ptr(), qt.data_ptr(), _ptr(w), sb, sv, sq, out.data_ptr(), B, V, Q, D, k, _stream()),
            "cti_bi_pool_fwd")
    return out


def bi_pool_shift(vt, qt, qadd, w):
    """out[b,d] = sum_vq vt[b,v,d] w[b,v,q] relu(qt[b,q,d] + qadd[b,d])  (k = 1; qadd (B,D) or None = 0), or None when no kernel forms the shifted
    operand on load at this shape (the caller materialises it).  Inference only."""
    vt16 = isinstance(vt, torch.Tensor) and vt.dtype == torch.bfloat16          # (round 5) bf16 rows from the hoisted projection: half the bytes of the streamed operand
    _req(vt, "vt", torch.bfloat16 if vt16 else torch.float32); _req(qt, "qt"); _req(w, "w")
    B, V, D = vt.shape
    Q = qt.shape[1]
    vt, qt = vt.contiguous(), qt.contiguous()
    if tuple(w.shape) != (B, V, Q) or B == 0 or V * Q == 0:
        return None
    if qadd is not None:
        _req(qadd, "qadd")
        qadd = qadd.contiguous()
    sb, sv, sq = w.stride()
    out = torch.empty((B, D), device=vt.device, dtype=torch.float32)
    lib = L.lib()
    if vt16:
        rc = lib.cti_bi_pool_shift_vt16_fwd(vt.data_ptr(), qt.data_ptr(), _ptr(qadd), w.data_ptr(), sb, sv, sq, out.data_ptr(), B, V, Q, D, _stream())
        if rc != L.E_UNSUPPORTED:
            L.check(rc, "cti_bi_pool_shift_vt16_fwd")
            return out
        vt = widen_bf16(vt)                                            # no bf16-reading kernel at this shape
    rc = lib.cti_bi_pool_shift_fwd(vt.data_ptr(), qt.data_ptr(), _ptr(qadd), w.data_ptr(), sb, sv, sq, out.data_ptr(), B, V, Q, D, _stream())
    if rc == L.E_UNSUPPORTED:
        return None
    L.check(rc, "cti_bi_pool_shift_fwd")
    return out


def bi_pool_shift_multi(vt, qt, adds, w, out):
    """out[b, :D] = sum_vq vt[b,v,:] w[b,v,q] relu(qt[b,q,:] + sum_i add_i[b,:])  with adds = [(device address, row stride in floats), ...] -- row stride 0 = one
    row for the whole batch -- summed as the pool loads them (cti_bi_pool_shift_multi_fwd: the raw split-K slabs of the products that feed the shift, no reduce
    launch in between).  out: a (B, D) view whose row stride may exceed D.  vt fp32 or bf16 rows.  False when no kernel takes the shape.  Inference only."""
    import ctypes as _C
    vt16 = vt.dtype == torch.bfloat16
    _req(vt, "vt", torch.bfloat16 if vt16 else torch.float32); _req(qt, "qt"); _req(w, "w"); _req(out, "out")
    B, V, D = vt.shape
    Q = qt.shape[1]
    vt, qt = vt.contiguous(), qt.contiguous()
    if tuple(w.shape) != (B, V, Q) or tuple(out.shape) != (B, D) or out.stride(1) != 1 or B == 0 or V * Q == 0 or len(adds) > 32:
        return False
    n = len(adds)
    ptrs = (_C.c_void_p * max(1, n))(*[int(a) for a, _ in adds])
    lds = (_C.c_int64 * max(1, n))(*[int(l) for _, l in adds])
    sb, sv, sq = w.stride()
    rc = L.lib().cti_bi_pool_shift_multi_fwd(vt.data_ptr(), 1 if vt16 else 0, qt.data_ptr(), ptrs, lds, n, w.data_ptr(), sb, sv, sq, out.data_ptr(), out.stride(0),
                                             B, V, Q, D, _stream())
    if rc == L.E_UNSUPPORTED:
        return False
    L.check(rc, "cti_bi_pool_shift_multi_fwd")
    return True


def gemm_pb_partials(x, w_planes, N, prec=None):
    """The raw fp32 split-K slabs (S, M, N) of x (M, K; any row stride) @ W^T against resident planes of an (N, K) weight -- no reduce pass, scale or bias
    (cti_gemm_pb_partials); bf16 modes only."""
    _req(x, "x")
    if x.dim() != 2 or x.stride(1) != 1:
        x = x.reshape(-1, x.shape[-1]).contiguous()
    M, K = x.shape
    pr = _prec(prec)
    lib = L.lib()
    S = lib.cti_gemm_pb_partials_count(M, int(N), K)
    out = torch.empty((S, M, int(N)), device=x.device, dtype=torch.float32)
    wsb = lib.cti_gemm_pb_partials_workspace_bytes(M, K, pr)
    ws = torch.empty(max(wsb, 16), device=x.device, dtype=torch.uint8)
    L.check(lib.cti_gemm_pb_partials(x.data_ptr(), x.stride(0), w_planes.data_ptr(), M, int(N), K, pr, out.data_ptr(), out.numel() * 4, ws.data_ptr(), wsb, _stream()),
            "cti_gemm_pb_partials")
    return out


def widen_bf16(x):
    """bf16 -> fp32 copy (off the fast path: a consumer without a bf16-reading kernel at its shape)."""
    return x.float()


def tri_pool_shift(vt, qt, at, qadd, aadd, w, v_rep=1):
    """out[b,d] = sum_vqa vt[b / v_rep, v, d] w[b,v,q,a] relu(qt[b,q,d] + qadd[b,d]) relu(at[b,a,d] + aadd[b,d])  (qadd / aadd (B,D) or None = 0), or
    None when no kernel forms the shifted operands on load at this shape.  Inference only."""
    vt16 = isinstance(vt, torch.Tensor) and vt.dtype == torch.bfloat16          # (round 5) bf16 rows from the hoisted projection
    _req(vt, "vt", torch.bfloat16 if vt16 else torch.float32)
    for t, n in ((qt, "qt"), (at, "at"), (w, "w")):
        _req(t, n)
    v_rep = int(v_rep)
    B, (V, D) = qt.shape[0], vt.shape[1:]
    Q, A = qt.shape[1], at.shape[1]
    if vt.shape[0] * v_rep != B or tuple(w.shape) != (B, V, Q, A) or B == 0 or V * Q * A == 0:
        return None
    vt, qt, at = vt.contiguous(), qt.contiguous(), at.contiguous()
    qadd = None if qadd is None else qadd.contiguous()
    aadd = None if aadd is None else aadd.contiguous()
    out = torch.empty((B, D), device=vt.device, dtype=torch.float32)
    sb, sv, sq, sa = w.stride()
    lib = L.lib()
    use_mfma = 0 if (get_precision() == "fp32" or _os.environ.get("CTI_NO_TRI_POOL_MFMA", "0") == "1") else (2 if get_precision() == "bf16" else 1)
    if vt16:
        rc = lib.cti_tri_pool_shift_vt16_fwd(vt.data_ptr(), qt.data_ptr(), at.data_ptr(), _ptr(qadd), _ptr(aadd), w.data_ptr(), sb, sv, sq, sa, out.data_ptr(),
                                             B, V, Q, A, D, v_rep, use_mfma, _stream())
        if rc != L.E_UNSUPPORTED:
            L.check(rc, "cti_tri_pool_shift_vt16_fwd")
            return out
        vt = widen_bf16(vt)
    rc = lib.cti_tri_pool_shift_fwd(vt.data_ptr(), qt.data_ptr(), at.data_ptr(), _ptr(qadd), _ptr(aadd), w.data_ptr(), sb, sv, sq, sa, out.data_ptr(),
                                    B, V, Q, A, D, v_rep, use_mfma, _stream())
    if rc == L.E_UNSUPPORTED and v_rep > 1:
        rc = lib.cti_tri_pool_shift_fwd(vt.repeat_interleave(v_rep, 0).data_ptr(), qt.data_ptr(), at.data_ptr(), _ptr(qadd), _ptr(aadd), w.data_ptr(), sb, sv, sq, sa,
                                        out.data_ptr(), B, V, Q, A, D, 1, use_mfma, _stream())
    if rc == L.E_UNSUPPORTED:
        return None
    L.check(rc, "cti_tri_pool_shift_fwd")
    return out


def rows_equal_prev(x):
    """(B,) uint8: row b of x (B, ...) holds the same bits as row b - 1 (element 0 is 0), or None when the rows are not 16-B multiples.  Any dtype (a byte compare)."""
    _req(x, "x", x.dtype if isinstance(x, torch.Tensor) else torch.float32)
    xc = x.contiguous()
    B = xc.shape[0]
    eq = torch.empty(B, device=x.device, dtype=torch.uint8)
    if B == 0:
        return eq
    rc = L.lib().cti_rows_equal_prev(xc.data_ptr(), xc[0].numel() * xc.element_size(), B, eq.data_ptr(), _stream())
    if rc == L.E_UNSUPPORTED:
        return None
    L.check(rc, "cti_rows_equal_prev")
    return eq


def replication_of(eq):
    """Largest r dividing the batch such that every row b with b % r != 0 equals its predecessor (HOST side: reads eq back -- one synchronisation
    of the stream that computed it)."""
    import numpy as np
    e = eq.cpu().numpy().astype(bool)
    B = e.shape[0]
    idx = np.arange(B)
    for r in sorted((d for d in range(1, B + 1) if B % d == 0), reverse=True):
        if e[idx % r != 0].all():
            return r
    return 1


def poison_unless_replicated(eq, r, out):
    """out.fill_(nan) on the device unless eq (rows_equal_prev of the batch) confirms groups of r identical rows."""
    L.check(L.lib().cti_poison_unless_replicated(eq.data_ptr(), eq.shape[0], int(r), out.data_ptr(), out.numel(), _stream()), "cti_poison_unless_replicated")
    return out


def joint_sums(q, cq, a=None, ca=0.0, Dq=None, dq=0.0, Da=None, da=0.0):
    """cq * q.sum(1) + ca * a.sum(1) + dq * Dq + da * Da  -> (B,H); q (B,Lq,H), a (B,La,H) or None, Dq / Da (B,H) or None: one pass."""
    _req(q, "q")
    q = q.contiguous()
    B, Lq, H = q.shape
    a = None if a is None else a.contiguous()
    Dq = None if Dq is None else Dq.contiguous()
    Da = None if Da is None else Da.contiguous()
    out = torch.empty((B, H), device=q.device, dtype=torch.float32)
    L.check(L.lib().cti_joint_sums(q.data_ptr(), Lq, float(cq), _ptr(a), 0 if a is None else a.shape[1], float(ca), _ptr(Dq), float(dq), _ptr(Da), float(da),
                                   out.data_ptr(), B, H, _stream()), "cti_joint_sums")
    return out


def axpby(x, a, y, b, out=None):
    """a * x + b * y (same shape, fp32)."""
    _req(x, "x"); _req(y, "y")
    x, y = x.contiguous(), y.contiguous()
    if out is None:
        out = torch.empty_like(x)
    L.check(L.lib().cti_axpby(x.data_ptr(), float(a), y.data_ptr(), float(b), out.data_ptr(), x.numel(), _stream()), "cti_axpby")
    return out


def bi_logits(vt, qt, h, h_scale, h_bias):
    """logits[b,g,v,q] = h_scale * sum_d vt[b,v,d] h[g,d] qt[b,q,d] + h_bias[g]."""
    vt16 = isinstance(vt, torch.Tensor) and vt.dtype == torch.bfloat16
    _req(vt, "vt", torch.bfloat16 if vt16 else torch.float32); _req(qt, "qt"); _req(h, "h")
    B, V, D = vt.shape
    Q = qt.shape[1]
    G = h.shape[0]
    vt, qt, h = vt.contiguous(), qt.contiguous(), h.contiguous()
    hb = h_bias.contiguous().view(-1) if h_bias is not None else None
    out = torch.empty((B, G, V, Q), device=vt.device, dtype=torch.float32)
    if out.numel() == 0:
        return out
    lib = L.lib()
    if vt16:
        rc = lib.cti_bi_logits_prec_vt16_fwd(vt.data_ptr(), qt.data_ptr(), h.data_ptr(), _ptr(h_scale), _ptr(hb), out.data_ptr(), B, G, V, Q, D, _prec(None), _stream())
        if rc != L.E_UNSUPPORTED:
            L.check(rc, "cti_bi_logits_prec_vt16_fwd")
            return out
        vt = widen_bf16(vt)
    if get_precision() != "fp32":                       # MFMA form (fp32-grade: three bf16 products; plain-bf16 mode: one); the exact-fp32 mode keeps the fp32 VALU kernel
        rc = lib.cti_bi_logits_prec_fwd(vt.data_ptr(), qt.data_ptr(), h.data_ptr(), _ptr(h_scale), _ptr(hb), out.data_ptr(), B, G, V, Q, D,
                                        _prec(None), _stream())
        if rc != L.E_UNSUPPORTED:
            L.check(rc, "cti_bi_logits_prec_fwd")
            return out
    L.check(lib.cti_bi_logits_fwd(vt.data_ptr(), qt.data_ptr(), h.data_ptr(), _ptr(h_scale), _ptr(hb), out.data_ptr(), B, G, V,
                                  Q, D, _stream()), "cti_bi_logits_fwd")
    return out


_bi_counters = {}


def biattention_forward(vt, qt, h, h_scale, h_bias, mask):
    """(p, logits) of BiAttention.forward_all from the projections: bilinear logits, -inf on the rows of `mask` ((B, V) uint8 or None) and the softmax over (V, Q)
    per glimpse, in ONE launch (cti_biattention_fwd).  Shapes outside that kernel: the separate calls."""
    vt16 = isinstance(vt, torch.Tensor) and vt.dtype == torch.bfloat16
    _req(vt, "vt", torch.bfloat16 if vt16 else torch.float32); _req(qt, "qt"); _req(h, "h")
    if vt16 and _os.environ.get("CTI_BIATT_FUSED", "0") == "1":
        vt = widen_bf16(vt)                                            # (the one-launch form reads fp32 rows)
    B, V, D = vt.shape
    Q = qt.shape[1]
    G = h.shape[0]
    # OFF by default: measured SLOWER than the two launches it replaces (tools/bench_pools.py, B = 256, G = 8, D = 3 072: 145 against 108 us) -- the release
    # fence in front of the per-sample counter makes every workgroup wait for its 4 000 atomic adds to be acknowledged by the L2, which a kernel boundary
    # overlaps with the next launch.  CTI_BIATT_FUSED=1 enables it (tests do).
    if B * G * V * Q > 0 and get_precision() != "fp32" and _os.environ.get("CTI_BIATT_FUSED", "0") == "1":
        vt, qt, h = vt.contiguous(), qt.contiguous(), h.contiguous()
        hb = h_bias.contiguous().view(-1) if h_bias is not None else None
        if mask is not None:
            _req(mask, "mask", torch.uint8)
            mask = mask.contiguous()
        dev = vt.device
        key = (dev.index if dev.index is not None else torch.cuda.current_device(), _stream())       # per stream: two streams must not share counters
        cnt = _bi_counters.get(key)
        if cnt is None or cnt.numel() < B:
            cnt = _bi_counters[key] = torch.zeros(max(B, 4096), device=dev, dtype=torch.int32)      # zero at entry, zeroed again by the kernel's last arrivers
        logits = torch.empty((B, G, V, Q), device=dev, dtype=torch.float32)
        p = torch.empty_like(logits)
        rc = L.lib().cti_biattention_fwd(vt.data_ptr(), qt.data_ptr(), h.data_ptr(), _ptr(h_scale), _ptr(hb), _ptr(mask), logits.data_ptr(), p.data_ptr(),
                                         cnt.data_ptr(), B, G, V, Q, D, _stream())
        if rc != L.E_UNSUPPORTED:
            L.check(rc, "cti_biattention_fwd")
            return p, logits
    logits = bi_logits(vt, qt, h, h_scale, h_bias)
    return masked_softmax_bi_(logits, mask), logits


# ---- backward-pass primitives ---------------------------------------------------------------------------------------
def gemm_nt(A, B, nb1=1, rA1=0, rB1=0, M=None, N=None, out=None, c_strides=None, sC1=0, scale=None, scale_div=1, bias=None, relu=False,
            prec=None, scale_bs=0, bias_bs=0, B_planes=None):
    """C[z][m,n] = act(scale * sum_k A[z*rA1 + m, k] * B[z*rB1 + n, k] + bias).  A (rowsA, K), B (rowsB, K) 2-D row-major."""
    _req(A, "A"); _req(B, "B")
    A2, lda = _rows2d(A); B2, ldb = _rows2d(B)
    K = A2.shape[1]
    assert B2.shape[1] == K
    M = A2.shape[0] if M is None else M
    N = B2.shape[0] if N is None else N
    if out is None:
        out = torch.empty((nb1, M, N) if nb1 > 1 else (M, N), device=A.device, dtype=torch.float32)
        ldc_m, ldc_n, sC1 = N, 1, M * N
    else:
        ldc_m, ldc_n = c_strides
    pr = _prec(prec)
    lib = L.lib()
    if B_planes is not None and pr != L.PREC_F32 and ldc_n == 1:
        wsb = lib.cti_gemm_nt_pb_workspace_bytes2(A2.shape[0], B2.shape[0], K, pr, int(nb1), int(M), int(N))
        ws = torch.empty(wsb, device=A.device, dtype=torch.uint8)
        L.check(lib.cti_gemm_nt_pb(A2.data_ptr(), lda, A2.shape[0], rA1, B_planes.data_ptr(), B2.shape[0], rB1, out.data_ptr(), ldc_m, sC1, nb1, M, N, K,
                                   _ptr(scale), int(scale_div), int(scale_bs), _ptr(bias), int(bias_bs), L.ACT_RELU if relu else L.ACT_NONE, pr,
                                   ws.data_ptr(), wsb, _stream()), "cti_gemm_nt_pb")
        return out
    wsb = lib.cti_gemm_nt_workspace_bytes(A2.shape[0], B2.shape[0], K, pr)
    ws = torch.empty(wsb, device=A.device, dtype=torch.uint8) if wsb else None
    L.check(lib.cti_gemm_nt(A2.data_ptr(), lda, A2.shape[0], rA1, 0, B2.data_ptr(), ldb, B2.shape[0], rB1, 0, out.data_ptr(), ldc_m, ldc_n,
                            sC1, 0, nb1, 1, M, N, K, _ptr(scale), int(scale_div), int(scale_bs), _ptr(bias), int(bias_bs),
                            L.ACT_RELU if relu else L.ACT_NONE, pr, _ptr(ws), wsb, _stream()), "cti_gemm_nt")
    return out


# Stream-K workspace of cti_gemm_bf16_rows_sk: one per (device, stream) -- two calls that may run at the same time must not share one -- zeroed once
# (the kernel leaves its flag words at zero).  64 MiB + 4 KiB each; built on first use by a product that is actually cut.
_sk_ws = {}
_sk_ws_bytes = [None]


def _sk_workspace(device, stream):
    key = (device.index if device.index is not None else torch.cuda.current_device(), stream)
    ws = _sk_ws.get(key)
    if ws is None:
        if _sk_ws_bytes[0] is None:
            _sk_ws_bytes[0] = int(L.lib().cti_gemm_bf16_rows_sk_workspace_bytes())
        ws = torch.empty(_sk_ws_bytes[0], device=device, dtype=torch.uint8)
        ws[:4096].zero_()                 # the flag page; the partial slots are written before they are read
        if not torch.cuda.is_current_stream_capturing():
            _sk_ws[key] = ws              # (a workspace born under capture belongs to that graph's pool: used by this call only, its zero-fill replays with the graph)
    return ws


def gemm16_sk_state(device=None):
    """(error words set, flag words left non-zero) over every stream-K workspace of `device` (tests; synchronises).  Both are 0 after any completed call."""
    err = left = 0
    for (d, _), ws in _sk_ws.items():
        if device is None or d == (device.index if device.index is not None else torch.cuda.current_device()):
            w = ws[:4096].view(torch.int32).cpu()
            n = torch.cuda.get_device_properties(d).multi_processor_count
            err += int(w[n] != 0); left += int((w[:n] != 0).sum())
    return err, left


_n_cu = {}


def _sk_cut(tiles, K):
    """Does csrc/cti_gemm16.hip's g16_sk_plan cut this product (256 x 256 tiles on all compute units)?  Only decides whether a workspace is worth building:
    by default three or more rounds of tiles that do not end on a whole round; with the test override (tuning(gemm16_sk=1)) any incomplete round."""
    d = torch.cuda.current_device()
    P = _n_cu.get(d)
    if P is None:
        P = _n_cu[d] = torch.cuda.get_device_properties(d).multi_processor_count
    if tiles <= P or tiles % P == 0 or K < 512:
        return False
    return tiles >= 3 * P or L.lib().cti_get_tuning(L.TUNE_GEMM16_SK) == 1


def gemm_bf16_rows(A, B_planes, rowsB, nb1=1, rA1=0, rB1=0, M=None, N=None, out_dtype=torch.float32, scale=None, scale_div=1, scale_bs=0, bias=None,
                   bias_bs=0, relu=False, stream_k=True):
    """C[z][m,n] = act(scale * sum_k A[z*rA1 + m, k] * W[z*rB1 + n, k] + bias) in the plain-bf16 arithmetic with A a row-major torch.bfloat16
    matrix read as it stands (no split pass) and W = split_operand(..., prec='bf16') planes of a (rowsB, K) weight; out_dtype float32 or bfloat16
    (bf16 rows = the next layer's A operand).  K % 32 == 0.  stream_k (round 6): products whose tiles do not fill whole rounds of the compute units
    are cut stream-K through cti_gemm_bf16_rows_sk (bit-identical results; a per-stream workspace); False = cti_gemm_bf16_rows, every tile whole."""
    _req(A, "A", torch.bfloat16)
    A2 = A.reshape(-1, A.shape[-1])
    if A2.stride(1) != 1:
        A2 = A2.contiguous()
    K = A2.shape[1]
    M = A2.shape[0] if M is None else M
    N = rowsB if N is None else N
    out = torch.empty((nb1, M, N) if nb1 > 1 else (M, N), device=A.device, dtype=out_dtype)
    st = _stream()
    if stream_k and _sk_cut(nb1 * ((M + 255) // 256) * ((N + 255) // 256), K):
        ws = _sk_workspace(A.device, st)
    else:
        ws = None
    if ws is not None:
        L.check(L.lib().cti_gemm_bf16_rows_sk(A2.data_ptr(), A2.stride(0), A2.shape[0], rA1, B_planes.data_ptr(), rowsB, rB1, out.data_ptr(),
                                              1 if out_dtype == torch.bfloat16 else 0, N, M * N, nb1, M, N, K, _ptr(scale), int(scale_div), int(scale_bs),
                                              _ptr(bias), int(bias_bs), L.ACT_RELU if relu else L.ACT_NONE, ws.data_ptr(), ws.numel(), st), "cti_gemm_bf16_rows_sk")
        return out
    L.check(L.lib().cti_gemm_bf16_rows(A2.data_ptr(), A2.stride(0), A2.shape[0], rA1, B_planes.data_ptr(), rowsB, rB1, out.data_ptr(),
                                       1 if out_dtype == torch.bfloat16 else 0, N, M * N, nb1, M, N, K, _ptr(scale), int(scale_div), int(scale_bs),
                                       _ptr(bias), int(bias_bs), L.ACT_RELU if relu else L.ACT_NONE, st), "cti_gemm_bf16_rows")
    return out


def split_operand(w, prec=None):
    """A (rows, K) fp32 matrix -> its resident bf16 hi/lo operand planes (one uint8 block) for gemm_nt(..., B_planes=...): split a weight
    once, multiply against it many times.  Returns None in the exact-fp32 mode (no planes there)."""
    _req(w, "w")
    pr = _prec(prec)
    if pr == L.PREC_F32:
        return None
    w2, ld = _rows2d(w)
    rows, K = w2.shape
    lib = L.lib()
    nb = lib.cti_operand_planes_bytes(rows, K)
    block = torch.empty(nb, device=w.device, dtype=torch.uint8)
    L.check(lib.cti_split_operand(w2.data_ptr(), ld, rows, K, block.data_ptr(), nb, _stream()), "cti_split_operand")
    return block


def quantize_f16f6(x, batch_rows=0, row_scale=None, scale_div=1):
    """(rows, K) fp32 -> the f16f6 operand planes (one uint8 block): f16 hi plane + block-scaled fp6 codes of the hi part and of the residual.
    batch_rows > 0: every batch of batch_rows rows starts at a multiple of 8 plane rows (for batched products).  row_scale: row m is
    multiplied by row_scale[m // scale_div] on the way in (a weight-normalised layer's g / ||V||)."""
    _req(x, "x")
    x2, ld = _rows2d(x)
    rows, K = x2.shape
    lib = L.lib()
    nb = lib.cti_f16f6_planes_bytes(rows, K, int(batch_rows))
    block = torch.empty(nb, device=x.device, dtype=torch.uint8)
    if row_scale is not None:
        L.check(lib.cti_quantize_f16f6_scaled(x2.data_ptr(), ld, rows, K, int(batch_rows), row_scale.data_ptr(), int(scale_div), block.data_ptr(), nb, _stream()),
                "cti_quantize_f16f6_scaled")
        return block
    L.check(lib.cti_quantize_f16f6(x2.data_ptr(), ld, rows, K, int(batch_rows), block.data_ptr(), nb, _stream()), "cti_quantize_f16f6")
    return block


def gemm_nt_f16f6(A, B, nb=1, M=None, N=None, gdiv=1, scale=None, scale_div=1, bias=None, relu=False):
    """C[z] = act(scale * A[z] @ B[z].T + bias) through the f16 + fp6 split product.  A (nb*M, K), B (nb*N, K) fp32; gdiv > 1: the rows of A are
    (m, g) pairs and C comes out as (nb, M/gdiv, N, gdiv) -- the mode-3 product."""
    _req(A, "A"); _req(B, "B")
    A2, _ = _rows2d(A); B2, _ = _rows2d(B)
    K = A2.shape[1]
    M = A2.shape[0] // nb if M is None else M
    N = B2.shape[0] // nb if N is None else N
    pa = quantize_f16f6(A2, M if nb > 1 else 0)
    pb = quantize_f16f6(B2, N if nb > 1 else 0)
    if gdiv > 1:
        C = torch.empty((nb, M // gdiv, N, gdiv), device=A.device, dtype=torch.float32)
        ldc_m, ldc_n, sC = N * gdiv, gdiv, M * N
    else:
        C = torch.empty((nb, M, N), device=A.device, dtype=torch.float32)
        ldc_m, ldc_n, sC = N, 1, M * N
    with _timed("gemm_nt_f16f6"):
        L.check(L.lib().cti_gemm_nt_f16f6(pa.data_ptr(), A2.shape[0], M if nb > 1 else 0, pb.data_ptr(), B2.shape[0], N if nb > 1 else 0, C.data_ptr(),
                                          ldc_m, ldc_n, sC, gdiv, nb, M, N, K, _ptr(scale), int(scale_div), _ptr(bias),
                                          L.ACT_RELU if relu else L.ACT_NONE, _stream()), "cti_gemm_nt_f16f6")
    return C


def linear_f16f6_planes(x_planes, rows, w_planes, M, K, batch_rows_out=0, bias=None, relu=False):
    """act(x @ w.T + bias) between f16f6 blocks: x (rows, K) and w (M, K) as blocks of quantize_f16f6 -> the block of the (rows, M) result.
    A weight-norm scale belongs in w's block (quantize_f16f6(w, row_scale=..., scale_div=...))."""
    lib = L.lib()
    nb = lib.cti_f16f6_planes_bytes(rows, M, int(batch_rows_out))
    y = torch.zeros(nb, device=x_planes.device, dtype=torch.uint8)
    with _timed("gemm_nt_f16f6_planes"):
        L.check(lib.cti_gemm_nt_f16f6_planes(w_planes.data_ptr(), M, x_planes.data_ptr(), rows, y.data_ptr(), nb, int(batch_rows_out), M, rows, K,
                                             _ptr(bias), L.ACT_RELU if relu else L.ACT_NONE, _stream()), "cti_gemm_nt_f16f6_planes")
    return y


def transpose(src, rows, cols, batch=1, s_src=0, ld_src=None, dst=None, s_dst=0, ld_dst=None):
    """dst[b][c][r] = src[b][r][c] for `batch` (rows x cols) matrices addressed with explicit strides."""
    ld_src = cols if ld_src is None else ld_src
    ld_dst = rows if ld_dst is None else ld_dst
    if dst is None:
        dst = torch.empty((batch, cols, rows), device=src.device, dtype=torch.float32)
        s_dst = cols * rows
    L.check(L.lib().cti_transpose_f32(src.data_ptr(), ld_src, s_src, dst.data_ptr(), ld_dst, s_dst, rows, cols, batch, _stream()),
            "cti_transpose_f32")
    return dst


def sum_batches(src, nb, n, alpha=1.0, out=None, beta=0.0):
    if out is None:
        out = torch.empty(n, device=src.device, dtype=torch.float32)
    L.check(L.lib().cti_sum_batches(src.data_ptr(), out.data_ptr(), nb, n, float(alpha), float(beta), _stream()), "cti_sum_batches")
    return out


def gemm_tn(a, b, prec=None):
    """a (M, N), b (M, K) contiguous -> a^T b (N, K): the weight-gradient contraction over the ROW axis, as a split-K batched NT GEMM:
    both operands are transposed chunk-wise ((S, Mc, .) -> (S, ., Mc)), S partial products are summed."""
    _req(a, "a"); _req(b, "b")
    M, N = a.shape
    K = b.shape[1]
    assert b.shape[0] == M and a.is_contiguous() and b.is_contiguous()
    pr = _prec(prec)
    if pr != L.PREC_F32 and M > 0 and _os.environ.get("CTI_GEMM_TN_LEGACY", "0") != "1":
        # straight to transposed operand planes + split-K over the rows, all behind one C-ABI call
        out = torch.empty((N, K), device=a.device, dtype=torch.float32)
        lib = L.lib()
        wsb = lib.cti_gemm_tn_workspace_bytes(M, N, K, pr)
        ws = torch.empty(wsb, device=a.device, dtype=torch.uint8)
        L.check(lib.cti_gemm_tn(a.data_ptr(), N, b.data_ptr(), K, out.data_ptr(), M, N, K, pr, ws.data_ptr(), wsb, _stream()), "cti_gemm_tn")
        return out
    S = max(1, min(128, M // 2048))
    Mc = (M + S - 1) // S
    Mc = (Mc + 31) // 32 * 32
    S = (M + Mc - 1) // Mc
    full, tail = M // Mc, M % Mc
    alloc = torch.zeros if tail else torch.empty
    aT = alloc((S, N, Mc), device=a.device, dtype=torch.float32)
    bT = alloc((S, K, Mc), device=a.device, dtype=torch.float32)
    if full:
        transpose(a, Mc, N, full, Mc * N, N, aT, N * Mc, Mc)
        transpose(b, Mc, K, full, Mc * K, K, bT, K * Mc, Mc)
    if tail:
        transpose(a[full * Mc:], tail, N, 1, 0, N, aT[full], 0, Mc)
        transpose(b[full * Mc:], tail, K, 1, 0, K, bT[full], 0, Mc)
    part = gemm_nt(aT.view(S * N, Mc), bT.view(S * K, Mc), nb1=S, rA1=N, rB1=K, M=N, N=K, prec=prec)
    if S == 1:
        return part.view(N, K)
    return sum_batches(part, S, N * K).view(N, K)


def gemm_nn(a, b, prec=None):
    """a (rows, N) @ b (N, K), both contiguous: the input gradient of a Linear layer.  bf16 modes: b goes straight to the planes of b^T
    (cti_gemm_nn); exact-fp32 mode: transposed fp32 copy + gemm_nt."""
    _req(a, "a"); _req(b, "b")
    rows, N = a.shape
    K = b.shape[1]
    assert b.shape[0] == N and a.is_contiguous() and b.is_contiguous()
    pr = _prec(prec)
    if pr == L.PREC_F32 or rows == 0:
        return gemm_nt(a, transpose(b, N, K).view(K, N), prec=prec)
    out = torch.empty((rows, K), device=a.device, dtype=torch.float32)
    lib = L.lib()
    wsb = lib.cti_gemm_nn_workspace_bytes(rows, N, K, pr)
    ws = torch.empty(wsb, device=a.device, dtype=torch.uint8)
    L.check(lib.cti_gemm_nn(a.data_ptr(), N, b.data_ptr(), K, out.data_ptr(), rows, N, K, pr, ws.data_ptr(), wsb, _stream()), "cti_gemm_nn")
    return out


def act_bwd(dy, y, scale, scale_div, relu):
    """dzs = scale[col // div] * dy * (y > 0); dbias = column sums of dy * (y > 0)."""
    _req(dy, "dy"); _req(y, "y")
    n = y.shape[-1]
    dy2 = dy.contiguous().view(-1, n); y2 = y.contiguous().view(-1, n)
    rows = y2.shape[0]
    dzs = torch.empty_like(y2)
    db = torch.empty(n, device=y.device, dtype=torch.float32)
    lib = L.lib()
    wsb = lib.cti_act_bwd_workspace_bytes(rows, n)
    ws = torch.empty(wsb, device=y.device, dtype=torch.uint8)
    L.check(lib.cti_act_bwd(dy2.data_ptr(), y2.data_ptr(), _ptr(scale), int(scale_div), dzs.data_ptr(), db.data_ptr(), rows, n,
                            L.ACT_RELU if relu else L.ACT_NONE, ws.data_ptr(), wsb, _stream()), "cti_act_bwd")
    return dzs, db


def wn_bwd(G, weight_v, weight_g, n_mats):
    G = G.contiguous(); wv = weight_v.contiguous(); g = weight_g.contiguous().view(-1)
    dV = torch.empty_like(wv)
    dg = torch.empty(n_mats, device=wv.device, dtype=torch.float32)
    lib = L.lib()
    wsb = lib.cti_wn_bwd_workspace_bytes(n_mats, wv.numel() // n_mats)
    ws = torch.empty(wsb, device=wv.device, dtype=torch.uint8) if wsb else None
    L.check(lib.cti_wn_bwd(G.data_ptr(), wv.data_ptr(), g.data_ptr(), dV.data_ptr(), dg.data_ptr(), n_mats, wv.numel() // n_mats,
                           _ptr(ws), wsb, _stream()), "cti_wn_bwd")
    return dV, dg


# ---- dropout streams -------------------------------------------------------------------------------------------------------
# Philox key of a call = f(torch.initial_seed(), host call counter) advanced on the DEVICE by a per-device step counter that
# FlatAdamaxDP.step() bumps (cti_counter_add): a training step captured in a hipGraph replays with the same kernel arguments and still
# draws fresh masks.  Re-seeding (torch.manual_seed) resets both counters, so a seed reproduces its masks regardless of what ran before;
# dropout_rng_state() / set_dropout_rng_state() carry them through a checkpoint.
_dropout_calls = [0]
_dropout_seed_seen = [None]
_rng_dev = {}


def _rng_tensor(device):
    key = device.index if device.index is not None else torch.cuda.current_device()
    t = _rng_dev.get(key)
    if t is None:
        t = _rng_dev[key] = torch.zeros(1, device=device, dtype=torch.int64)
    return t


def _dp_rank():
    try:
        import torch.distributed as dist
        return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
    except Exception:
        return 0


def _dropout_seed(device):
    seed0 = torch.initial_seed()
    if _dropout_seed_seen[0] != seed0:                       # torch.manual_seed(...) since the last call: restart the streams
        _dropout_seed_seen[0] = seed0
        _dropout_calls[0] = 0
        if not torch.cuda.is_current_stream_capturing():
            for t in _rng_dev.values():
                t.zero_()
    # key = seed0 * PHI + rank * C3 + call counter, advanced on the device by step * C2 (cti_dropout_g): three different odd multipliers, so
    # neither adjacent seeds (1204, 1205, ...) nor the ranks of a data-parallel run that all call torch.manual_seed(seed) share mask streams
    # shifted by a step / a call.
    seed = (seed0 * 0x9E3779B97F4A7C15 + _dp_rank() * 0xA0761D6478BD642F + _dropout_calls[0]) & 0xFFFFFFFFFFFFFFFF
    _dropout_calls[0] += 1
    return seed, _rng_tensor(device).data_ptr()


def rng_advance(device=None, inc=1):
    """Bump the device-side step counter of the dropout streams (stream-ordered; FlatAdamaxDP.step() calls it once per step)."""
    t = _rng_tensor(device if device is not None else torch.device("cuda", torch.cuda.current_device()))
    L.check(L.lib().cti_counter_add(t.data_ptr(), int(inc), _stream()), "cti_counter_add")


def dropout_rng_state():
    """(seed seen, host call counter, {device index: device step counter}) -- synchronises."""
    return {"seed": _dropout_seed_seen[0], "calls": _dropout_calls[0], "steps": {k: int(t.item()) for k, t in _rng_dev.items()}}


def set_dropout_rng_state(st):
    _dropout_seed_seen[0], _dropout_calls[0] = st["seed"], int(st["calls"])
    for k, v in st.get("steps", {}).items():
        _rng_tensor(torch.device("cuda", int(k))).fill_(int(v))


def reset_dropout_rng():
    """Test hook: restart the dropout streams as if torch.manual_seed had just been called."""
    _dropout_seed_seen[0] = None


def dropout(x, p, mask=None, copies=1):
    """Forward (mask=None): returns (y, mask) with a fresh Philox stream (see above); copies = R > 1 returns R independently masked copies of x
    stacked on a new leading axis.  Backward: pass the stored mask, returns dy * mask / (1-p) (same shape as dy)."""
    _req(x, "x")
    x = x.contiguous()
    if mask is None:
        shape = ((copies,) + tuple(x.shape)) if copies > 1 else tuple(x.shape)
        y = torch.empty(shape, device=x.device, dtype=torch.float32)
        mask = torch.empty(shape, device=x.device, dtype=torch.uint8)
        seed, rng = _dropout_seed(x.device)
        L.check(L.lib().cti_dropout_g(x.data_ptr(), y.data_ptr(), mask.data_ptr(), y.numel(), float(p), seed, 0, 0,
                                      x.numel() if copies > 1 else 0, rng, _stream()), "cti_dropout")
        return y, mask
    y = torch.empty_like(x)
    L.check(L.lib().cti_dropout(x.data_ptr(), y.data_ptr(), mask.data_ptr(), x.numel(), float(p), 0, 0, 1, 0, _stream()), "cti_dropout")
    return y


def dropout_mask(shape, p, device):
    """A fresh keep-mask (uint8, 1 = keep) from the same Philox stream as dropout(); nothing else is written."""
    mask = torch.empty(tuple(shape), device=device, dtype=torch.uint8)
    seed, rng = _dropout_seed(mask.device)
    L.check(L.lib().cti_dropout_g(None, None, mask.data_ptr(), mask.numel(), float(p), seed, 0, 0, 0, rng, _stream()), "cti_dropout (mask only)")
    return mask


def ranknets_drop_fwd(x2, mask, wv, scale, bias, R, p, relu):
    """Train-mode rank nets without the R masked copies (cti_ranknets_drop_fwd); None when the shape is outside the kernel."""
    rows, h = x2.shape
    hr = wv.shape[0] // R
    y = torch.empty((rows, R * hr), device=x2.device, dtype=torch.float32)
    pr = _prec(None)
    if pr != L.PREC_F32 and rows > 0 and _os.environ.get("CTI_NO_RANKNETS_MFMA", "0") != "1":       # bf16 split products on the matrix cores (h = 512, hr = 16)
        lib = L.lib()
        wsb = lib.cti_ranknets_drop_fwd_mfma_workspace_bytes(h, R, hr)
        ws = torch.empty(wsb, device=x2.device, dtype=torch.uint8)
        rc = lib.cti_ranknets_drop_fwd_mfma(_req(x2, "x").data_ptr(), mask.data_ptr(), _req(wv, "W").data_ptr(), _req(scale, "scale").data_ptr(), _ptr(bias),
                                            y.data_ptr(), rows, h, R, hr, float(p), 1 if relu else 0, L.PREC_BF16 if pr == L.PREC_BF16 else L.PREC_BF16X3,
                                            ws.data_ptr(), wsb, _stream())
        if rc != L.E_UNSUPPORTED:
            L.check(rc, "cti_ranknets_drop_fwd_mfma")
            return y
    rc = L.lib().cti_ranknets_drop_fwd(_req(x2, "x").data_ptr(), mask.data_ptr(), _req(wv, "W").data_ptr(), _req(scale, "scale").data_ptr(),
                                       _ptr(bias), y.data_ptr(), rows, h, R, hr, float(p), 1 if relu else 0, _stream())
    if rc == L.E_UNSUPPORTED:
        return None
    L.check(rc, "cti_ranknets_drop_fwd")
    return y


def ranknets_drop_dw(dzs, x2, mask, R, p):
    rows, h = x2.shape
    hr = dzs.shape[1] // R
    G = torch.empty((R * hr, h), device=x2.device, dtype=torch.float32)
    L.check(L.lib().cti_ranknets_drop_dw(_req(dzs, "dzs").data_ptr(), x2.data_ptr(), mask.data_ptr(), G.data_ptr(), rows, h, R, hr, float(p), _stream()),
            "cti_ranknets_drop_dw")
    return G


def ranknets_drop_dx(dzs, wv, mask, R, p):
    rows = dzs.shape[0]
    h = wv.shape[1]
    hr = wv.shape[0] // R
    dx = torch.empty((rows, h), device=dzs.device, dtype=torch.float32)
    L.check(L.lib().cti_ranknets_drop_dx(_req(dzs, "dzs").data_ptr(), _req(wv, "W").data_ptr(), mask.data_ptr(), dx.data_ptr(), rows, h, R, hr, float(p),
                                         _stream()), "cti_ranknets_drop_dx")
    return dx


def paralind_mbuild_bwd(dM, Vr, Qr, Teff, prec=None):
    """-> dVr, dQr, dTeff (summed over the batch)."""
    R, I, J, K, G = Teff.shape
    B, V, _ = Vr.shape
    Q = Qr.shape[1]
    if not (I == J == K):
        raise L.CtiError("M-build backward is built for cubic cores (hv = hq = ha)")
    dM, Vr, Qr, Teff = dM.contiguous(), Vr.contiguous(), Qr.contiguous(), Teff.contiguous()
    dVr, dQr = torch.empty_like(Vr), torch.empty_like(Qr)
    lib = L.lib()
    rc = L.E_UNSUPPORTED
    pr = _prec(prec)
    nparts = B
    if pr != L.PREC_F32 and B > 0 and _os.environ.get("CTI_NO_MBUILD_BWD_MFMA", "0") != "1":      # matrix-core form (fp32-grade split products); exact-fp32 mode keeps the VALU kernels
        nparts = lib.cti_paralind_mbuild_bwd_mfma_partials(B, R)                                   # one partial of dT_eff per chunk of samples
        part = torch.empty((nparts,) + tuple(Teff.shape), device=Vr.device, dtype=torch.float32)
        rc = lib.cti_paralind_mbuild_bwd_mfma(dM.data_ptr(), Vr.data_ptr(), Qr.data_ptr(), Teff.data_ptr(), dVr.data_ptr(), dQr.data_ptr(),
                                              part.data_ptr(), B, V, Q, R, I, G, L.PREC_BF16 if pr == L.PREC_BF16 else L.PREC_BF16X3, _stream())
    if rc == L.E_UNSUPPORTED:
        nparts = B
        part = torch.empty((B,) + tuple(Teff.shape), device=Vr.device, dtype=torch.float32)
        rc = lib.cti_paralind_mbuild_bwd(dM.data_ptr(), Vr.data_ptr(), Qr.data_ptr(), Teff.data_ptr(), dVr.data_ptr(), dQr.data_ptr(),
                                         part.data_ptr(), B, V, Q, R, I, G, _stream())
    if rc == L.E_UNSUPPORTED:                                  # h/rank outside {4, 8, 16} (or beyond the staged kernels' budgets): the generic VALU form
        wsb = lib.cti_paralind_mbuild_bwd_generic_workspace_bytes(B, V, R, I, G)
        ws = torch.empty(wsb, device=Vr.device, dtype=torch.uint8)
        rc = lib.cti_paralind_mbuild_bwd_generic(dM.data_ptr(), Vr.data_ptr(), Qr.data_ptr(), Teff.data_ptr(), dVr.data_ptr(), dQr.data_ptr(),
                                                 part.data_ptr(), B, V, Q, R, I, G, ws.data_ptr(), wsb, _stream())
    L.check(rc, "cti_paralind_mbuild_bwd")
    dT = sum_batches(part, nparts, Teff.numel()).view(Teff.shape)
    return dVr, dQr, dT


def paralind_core_bwd(dout, M, Ar, prec=None):
    """dout (B,V,Q,A,G), M (B,V,Q,G,K), Ar (B,A,K) -> dM, dAr (cti_paralind_core_bwd; shapes outside it: two batched NT GEMMs over
    transposed operands)."""
    B, V, Q, G, K = M.shape
    A = Ar.shape[1]
    VQG = V * Q * G
    dout, M, Ar = dout.contiguous(), M.contiguous(), Ar.contiguous()
    if B > 0:
        # two streaming passes (contraction of length A / A output rows): no transposed copies or plane splits of the (B, V*Q*G, K) tensors
        dM = torch.empty_like(M)
        dAr = torch.empty_like(Ar)
        rc = L.lib().cti_paralind_core_bwd(_req(dout, "dout").data_ptr(), _req(M, "M").data_ptr(), _req(Ar, "Ar").data_ptr(), dM.data_ptr(),
                                           dAr.data_ptr(), B, V, Q, A, G, K, _stream())
        if rc != L.E_UNSUPPORTED:
            L.check(rc, "cti_paralind_core_bwd")
            return dM, dAr
    doutT = transpose(dout, A, G, B * V * Q, A * G)                       # (B*VQ, G, A): rows (vq,g), a contiguous
    ArT = transpose(Ar, A, K, B, A * K)                                   # (B, K, A)
    dM = gemm_nt(doutT.view(B * VQG, A), ArT.view(B * K, A), nb1=B, rA1=VQG, rB1=K, M=VQG, N=K, prec=prec)
    dout2 = transpose(doutT, VQG, A, B, VQG * A)                          # (B, A, VQG)
    Mt = transpose(M, VQG, K, B, VQG * K)                                 # (B, K, VQG)
    dAr = gemm_nt(dout2.view(B * A, VQG), Mt.view(B * K, VQG), nb1=B, rA1=A, rB1=K, M=A, N=K, prec=prec)
    return dM.view(B, V, Q, G, K), dAr.view(B, A, K)


def masked_softmax_tri_bwd(p, dp):
    B, V, Q, A, G = p.shape
    p, dp = p.contiguous(), dp.contiguous()
    dl = torch.empty_like(p)
    lib = L.lib()
    wsb = lib.cti_softmax_tri_bwd_workspace_bytes(B, V, Q * A, G)
    ws = torch.empty(wsb, device=p.device, dtype=torch.uint8)
    L.check(lib.cti_masked_softmax_tri_bwd(p.data_ptr(), dp.data_ptr(), dl.data_ptr(), B, V, Q * A, G, ws.data_ptr(), wsb, _stream()),
            "cti_masked_softmax_tri_bwd")
    return dl


def masked_softmax_bi_bwd(p, dp):
    B, G, V, Q = p.shape
    p, dp = p.contiguous(), dp.contiguous()
    dl = torch.empty_like(p)
    L.check(L.lib().cti_masked_softmax_bi_bwd(p.data_ptr(), dp.data_ptr(), dl.data_ptr(), B * G, V * Q, _stream()), "cti_masked_softmax_bi_bwd")
    return dl


def tri_pool_bwd(dout, vt, qt, at, w, need_dw=True):
    B, V, D = vt.shape
    Q, A = qt.shape[1], at.shape[1]
    dout, vt, qt, at = dout.contiguous(), vt.contiguous(), qt.contiguous(), at.contiguous()
    dvt, dqt, dat = torch.empty_like(vt), torch.empty_like(qt), torch.empty_like(at)
    dw = torch.empty((B, V, Q, A), device=vt.device, dtype=torch.float32) if need_dw else None
    sb, sv, sq, sa = w.stride()
    lib = L.lib()
    dw_done = False
    if need_dw and get_precision() != "fp32":                 # the attention gradient as an MFMA contraction over the channels
        rc = lib.cti_pool_dw_mfma(dout.data_ptr(), vt.data_ptr(), qt.data_ptr(), at.data_ptr(), dw.data_ptr(), B, V, Q, A, D, _stream())
        if rc != L.E_UNSUPPORTED:
            L.check(rc, "cti_pool_dw_mfma")
            dw_done = True
    L.check(lib.cti_tri_pool_bwd(dout.data_ptr(), vt.data_ptr(), qt.data_ptr(), at.data_ptr(), w.data_ptr(), sb, sv, sq, sa, dvt.data_ptr(),
                                 dqt.data_ptr(), dat.data_ptr(), 0 if dw_done else _ptr(dw), B, V, Q, A, D, _stream()), "cti_tri_pool_bwd")
    return dvt, dqt, dat, dw


def bi_pool_bwd(dout, vt, qt, w, k, need_dw=True):
    B, V, D = vt.shape
    Q = qt.shape[1]
    dout, vt, qt = dout.contiguous(), vt.contiguous(), qt.contiguous()
    dvt, dqt = torch.empty_like(vt), torch.empty_like(qt)
    dw = torch.empty((B, V, Q), device=vt.device, dtype=torch.float32) if (need_dw and w is not None) else None
    sb, sv, sq = w.stride() if w is not None else (0, 0, 0)
    lib = L.lib()
    dw_done = False
    if dw is not None and k == 1 and get_precision() != "fp32":
        rc = lib.cti_pool_dw_mfma(dout.data_ptr(), vt.data_ptr(), qt.data_ptr(), 0, dw.data_ptr(), B, V, Q, 1, D, _stream())
        if rc != L.E_UNSUPPORTED:
            L.check(rc, "cti_pool_dw_mfma")
            dw_done = True
    L.check(lib.cti_bi_pool_bwd(dout.data_ptr(), vt.data_ptr(), qt.data_ptr(), _ptr(w), sb, sv, sq, dvt.data_ptr(), dqt.data_ptr(),
                                0 if dw_done else _ptr(dw), B, V, Q, D, k, _stream()), "cti_bi_pool_bwd")
    return dvt, dqt, dw


def bi_logits_bwd(dl, vt, qt, h, h_scale):
    """-> dvt, dqt, G_h (G,D) = h_scale * dL/d(h_scale*h) summed over the batch, dh_bias (G,)."""
    B, V, D = vt.shape
    Q = qt.shape[1]
    G = h.shape[0]
    dl, vt, qt, h = dl.contiguous(), vt.contiguous(), qt.contiguous(), h.contiguous()
    dvt, dqt = torch.empty_like(vt), torch.empty_like(qt)
    hp = torch.empty((B, G, D), device=vt.device, dtype=torch.float32)
    bp = torch.empty((B, G), device=vt.device, dtype=torch.float32)
    lib = L.lib()
    done = False
    if get_precision() != "fp32":                              # the three contractions on the MFMA (fp32-grade), the bias gradient as row sums
        rc = lib.cti_bi_logits_bwd_mfma(dl.data_ptr(), vt.data_ptr(), qt.data_ptr(), h.data_ptr(), _ptr(h_scale), dvt.data_ptr(), dqt.data_ptr(),
                                        hp.data_ptr(), B, G, V, Q, D, _stream())
        if rc != L.E_UNSUPPORTED:
            L.check(rc, "cti_bi_logits_bwd_mfma")
            L.check(lib.cti_row_sum(dl.data_ptr(), bp.data_ptr(), B * G, V * Q, _stream()), "cti_row_sum")
            done = True
    if not done:
        L.check(lib.cti_bi_logits_bwd(dl.data_ptr(), vt.data_ptr(), qt.data_ptr(), h.data_ptr(), _ptr(h_scale), dvt.data_ptr(), dqt.data_ptr(),
                                      hp.data_ptr(), bp.data_ptr(), B, G, V, Q, D, _stream()), "cti_bi_logits_bwd")
    return dvt, dqt, sum_batches(hp, B, G * D).view(G, D), sum_batches(bp, B, G)


# ---- rows either side of the CTI path: embedding, GRU, Swish, sequence sums, losses -------------------------
def embedding(tokens, table0, table1=None):
    """tokens (...,) int64 -> (..., dim) or (..., 2*dim) when a second table is concatenated."""
    _req(tokens, "tokens", torch.int64); _req(table0, "table0")
    tok = tokens.contiguous()
    rows, dim = table0.shape
    t0 = table0.contiguous()
    t1 = None
    if table1 is not None:
        _req(table1, "table1")
        if tuple(table1.shape) != (rows, dim):
            raise ValueError("the two embedding tables differ in shape: %s vs %s" % (tuple(table0.shape), tuple(table1.shape)))
        t1 = table1.contiguous()
    out = torch.empty(tuple(tok.shape) + ((2 if t1 is not None else 1) * dim,), device=tok.device, dtype=torch.float32)
    L.check(L.lib().cti_embedding_fwd(tok.data_ptr(), t0.data_ptr(), _ptr(t1), out.data_ptr(), tok.numel(), dim, rows, _stream()),
            "cti_embedding_fwd")
    return out


def embedding_rows16(tokens, table0, table1=None):
    """tokens (...,) int64 -> bf16 rows (..., ld), ld = the row's width rounded up to a multiple of 32 (zero beyond the width): the GRU's input-side operand as
    the plain-bf16 mode multiplies it (gru_forward takes it as it stands: no fp32 word vectors, no split pass)."""
    _req(tokens, "tokens", torch.int64); _req(table0, "table0")
    tok = tokens.contiguous()
    rows, dim = table0.shape
    t0 = table0.contiguous()
    t1 = None
    if table1 is not None:
        _req(table1, "table1")
        if tuple(table1.shape) != (rows, dim):
            raise ValueError("the two embedding tables differ in shape: %s vs %s" % (tuple(table0.shape), tuple(table1.shape)))
        t1 = table1.contiguous()
    width = (2 if t1 is not None else 1) * dim
    ld = (width + 31) // 32 * 32
    out = torch.empty(tuple(tok.shape) + (ld,), device=tok.device, dtype=torch.bfloat16)
    L.check(L.lib().cti_embedding_fwd_bf16(tok.data_ptr(), t0.data_ptr(), _ptr(t1), out.data_ptr(), ld, tok.numel(), dim, rows, _stream()), "cti_embedding_fwd_bf16")
    return out


def embedding_bwd(tokens, dout, col_off, rows, dim, padding_idx):
    """-> dtable (rows, dim) = scatter-add of dout[..., col_off:col_off+dim] by token (the padding row stays zero)."""
    tok = tokens.contiguous()
    d2 = dout.contiguous().view(tok.numel(), -1)
    dt = torch.zeros((rows, dim), device=dout.device, dtype=torch.float32)
    L.check(L.lib().cti_embedding_bwd(tok.data_ptr(), d2.data_ptr(), d2.shape[1], int(col_off), dt.data_ptr(), tok.numel(), dim, rows,
                                      int(padding_idx), _stream()), "cti_embedding_bwd")
    return dt


def col_sum(x2, alpha=1.0):
    """(rows, n) contiguous -> (n,) column sums (two-stage reduction over row groups)."""
    _req(x2, "x")
    rows, n = x2.shape
    out = torch.empty(n, device=x2.device, dtype=torch.float32)
    if rows == 0:
        return out.zero_()
    lib = L.lib()
    wsb = lib.cti_col_sum_workspace_bytes(rows, n)
    ws = torch.empty(wsb, device=x2.device, dtype=torch.uint8)
    L.check(lib.cti_col_sum(x2.data_ptr(), rows, n, out.data_ptr(), float(alpha), 0.0, ws.data_ptr(), wsb, _stream()), "cti_col_sum")
    return out


# The GRU's persistent form (cti_gru.hip: all steps in ONE launch whose workgroups meet at counters) needs every one of its workgroups resident, so two such
# launches must never share the device: callers that enable it keep their GRUs on ONE stream (base_model._TriModel), run_concurrently's sibling streams do not
# get it, and a process that drives one device from several host threads should leave it off.  CTI_GRU_PERSISTENT=1 enables it (default off: see DESIGN.md,
# round 6 -- beside the other streams' work the whole forward gains nothing from it).
use_persistent_gru = _os.environ.get("CTI_GRU_PERSISTENT", "0") == "1"


def gru_persistent_ok():
    """True when gru_forward may take the persistent form for an inference call in the plain-bf16 mode (the caller then keeps its GRUs on one stream)."""
    return use_persistent_gru and not _no_nested_fork[0] and get_precision() == "bf16" and not torch.is_grad_enabled()


def gru_forward(x, w_ih, w_hh, b_ih, b_hh, want_save=False, prec=None, w_planes=None):
    """One-layer, one-direction nn.GRU(batch_first=True) from a zero state: x (B,T,in) -> every hidden state (B,T,H), in ONE library
    call (the time loop lives behind the C ABI).  want_save: also returns save (T,B,5,H) = (r, z, n, W_hn h + b_hn, h_t)."""
    x16 = x.dtype == torch.bfloat16                                   # embedding_rows16's rows: (B, T, I rounded up to 32), the plain-bf16 mode only
    _req(x, "x", torch.bfloat16 if x16 else torch.float32)
    B, T, I = x.shape
    if x16:
        I = w_ih.shape[1]
        if _prec(prec) != L.PREC_BF16 or want_save or x.shape[2] != (I + 31) // 32 * 32:
            x, x16 = widen_bf16(x[:, :, :I]).contiguous(), False
    H = w_hh.shape[1]
    out = torch.empty((B, T, H), device=x.device, dtype=torch.float32)
    save = torch.empty((T, B, 5, H), device=x.device, dtype=torch.float32) if want_save else None
    if B * T == 0:
        return out, save
    pr = _prec(prec)
    lib = L.lib()
    wsb = lib.cti_gru_forward_workspace_bytes(B, T, I, H, pr)
    ws = torch.empty(wsb, device=x.device, dtype=torch.uint8)
    persistent = pr == L.PREC_BF16 and not want_save and gru_persistent_ok()
    if persistent:
        L.check(lib.cti_set_tuning(L.TUNE_GRU_PERSISTENT, 1), "cti_set_tuning")
    try:
        with _timed("gru_forward_%dx%dx%d->%d" % (B, T, I, H)):
            if x16:
                L.check(lib.cti_gru_forward_x16(x.contiguous().data_ptr(), x.shape[2], w_ih.contiguous().data_ptr(), w_hh.contiguous().data_ptr(),
                                                b_ih.contiguous().data_ptr(), b_hh.contiguous().data_ptr(), out.data_ptr(), 0, B, T, I, H, pr,
                                                _ptr(w_planes[0]) if w_planes else 0, _ptr(w_planes[1]) if w_planes else 0,
                                                ws.data_ptr(), wsb, _stream()), "cti_gru_forward_x16")
                return out, save
            L.check(lib.cti_gru_forward(x.contiguous().data_ptr(), w_ih.contiguous().data_ptr(), w_hh.contiguous().data_ptr(),
                                        b_ih.contiguous().data_ptr(), b_hh.contiguous().data_ptr(), out.data_ptr(), _ptr(save), B, T, I, H, pr,
                                        _ptr(w_planes[0]) if w_planes and pr != L.PREC_F32 else 0,
                                        _ptr(w_planes[1]) if w_planes and pr != L.PREC_F32 else 0,
                                        ws.data_ptr(), wsb, _stream()), "cti_gru_forward")
    finally:
        if persistent:
            L.check(lib.cti_set_tuning(L.TUNE_GRU_PERSISTENT, 0), "cti_set_tuning")
    return out, save


def gru_backward(dout, x, w_ih, w_hh, save, need_dx=True, prec=None):
    """-> dx (or None), dW_ih, dW_hh, db_ih, db_hh.  BPTT in one library call, then four weight-gradient contractions."""
    B, T, H = dout.shape
    I = x.shape[2]
    dev = dout.device
    dgi = torch.empty((B, T, 3 * H), device=dev, dtype=torch.float32)
    dgh = torch.empty((T, B, 3 * H), device=dev, dtype=torch.float32)
    pr = _prec(prec)
    lib = L.lib()
    wsb = lib.cti_gru_backward_workspace_bytes(B, T, H, pr)
    ws = torch.empty(wsb, device=dev, dtype=torch.uint8)
    L.check(lib.cti_gru_backward(dout.contiguous().data_ptr(), w_hh.contiguous().data_ptr(), save.data_ptr(), dgi.data_ptr(), dgh.data_ptr(),
                                 B, T, H, pr, ws.data_ptr(), wsb, _stream()), "cti_gru_backward")
    dgi2, dgh2 = dgi.view(B * T, 3 * H), dgh.view(T * B, 3 * H)
    if T > 1:
        h_prev = save[:-1, :, 4, :].contiguous().view((T - 1) * B, H)           # h_0 .. h_{T-2}, time-major like dgh[1:]
        dW_hh = gemm_tn(dgh2[B:], h_prev)
    else:
        dW_hh = torch.zeros((3 * H, H), device=dev, dtype=torch.float32)
    dW_ih = gemm_tn(dgi2, x.contiguous().view(B * T, I))
    db_hh = col_sum(dgh2)
    db_ih = col_sum(dgi2)
    dx = None
    if need_dx:
        dx = gemm_nn(dgi2, w_ih.contiguous()).view(B, T, I)
    return dx, dW_ih, dW_hh, db_ih, db_hh


def swish(x):
    _req(x, "x")
    xc = x.contiguous()
    y = torch.empty_like(xc)
    L.check(L.lib().cti_swish_fwd(xc.data_ptr(), y.data_ptr(), xc.numel(), _stream()), "cti_swish_fwd")
    return y


def swish_bwd(x, dy):
    xc, dyc = x.contiguous(), dy.contiguous()
    dx = torch.empty_like(xc)
    L.check(L.lib().cti_swish_bwd(xc.data_ptr(), dyc.data_ptr(), dx.data_ptr(), xc.numel(), _stream()), "cti_swish_bwd")
    return dx


def seq_sum(x, out=None, beta=0.0):
    """x (B,L,H) -> (B,H) = beta*out + sum over L."""
    _req(x, "x")
    B, Lq, H = x.shape
    xc = x.contiguous()
    if out is None:
        out = torch.empty((B, H), device=x.device, dtype=torch.float32)
        beta = 0.0
    L.check(L.lib().cti_seq_sum(xc.data_ptr(), out.data_ptr(), B, Lq, H, float(beta), _stream()), "cti_seq_sum")
    return out


def seq_bcast_add(x, y, Lq=None):
    """x (B,L,H) or None, y (B,H) -> x + y[:, None, :]  (x None: y broadcast to (B,Lq,H))."""
    _req(y, "y")
    B, H = y.shape
    if x is not None:
        _req(x, "x")
        Lq = x.shape[1]
        x = x.contiguous()
    out = torch.empty((B, Lq, H), device=y.device, dtype=torch.float32)
    L.check(L.lib().cti_seq_bcast_add(_ptr(x), y.contiguous().data_ptr(), out.data_ptr(), B, Lq, H, _stream()), "cti_seq_bcast_add")
    return out


def linear_residual(x, w_planes, scale, scale_div, bias, seq, acc=None, beta=0.0, prec=None):
    """out[b,l,:] = seq[b,l,:] + (scale * (x @ W^T) + bias)[b,:]; acc[b,:] = beta * acc[b,:] + sum_l out[b,l,:] when acc is given.  x (B,K),
    W as split_operand(weight_v) planes of an (N,K) weight, seq (B,L,N).  Returns out, or None when the fused form does not apply (exact-fp32
    mode, N % 4 != 0): the caller then takes wn_linear + seq_bcast_add (+ seq_sum)."""
    _req(x, "x"); _req(seq, "seq")
    pr = _prec(prec)
    B, L, N = seq.shape
    K = x.shape[-1]
    if pr == L_.PREC_F32 or w_planes is None or N % 4 or x.shape[0] != B or B == 0 or L == 0:
        return None
    x2, ldx = _rows2d(x)
    seq = seq.contiguous()
    out = torch.empty_like(seq)
    lib = L_.lib()
    wsb = lib.cti_linear_residual_workspace_bytes(B, N, K, pr)
    ws = torch.empty(wsb, device=x.device, dtype=torch.uint8)
    L_.check(lib.cti_linear_residual_pb(x2.data_ptr(), ldx, w_planes.data_ptr(), _ptr(scale), int(scale_div), _ptr(bias), seq.data_ptr(), out.data_ptr(),
                                        _ptr(acc), float(beta), B, L, N, K, pr, ws.data_ptr(), wsb, _stream()), "cti_linear_residual_pb")
    return out


def bce_logits_sum(x, target):
    """sum over every element of BCE-with-logits -> 0-d tensor."""
    _req(x, "x"); _req(target, "target")
    n = x.shape[-1]
    x2, t2 = x.contiguous().view(-1, n), target.contiguous().view(-1, n)
    rows = torch.empty(x2.shape[0], device=x.device, dtype=torch.float32)
    L.check(L.lib().cti_bce_logits_rows_fwd(x2.data_ptr(), t2.data_ptr(), rows.data_ptr(), x2.shape[0], n, _stream()), "cti_bce_logits_rows_fwd")
    return rows


def bce_logits_bwd(x, target, upstream, coef, dx=None, beta=0.0):
    xc, tc = x.contiguous(), target.contiguous()
    if dx is None:
        dx = torch.empty_like(xc)
        beta = 0.0
    L.check(L.lib().cti_bce_logits_bwd(xc.data_ptr(), tc.data_ptr(), _ptr(upstream), float(coef), dx.data_ptr(), xc.numel(), float(beta),
                                       _stream()), "cti_bce_logits_bwd")
    return dx


def kd_rows(x, knowledge, T):
    _req(x, "x"); _req(knowledge, "knowledge")
    n = x.shape[-1]
    x2, k2 = x.contiguous().view(-1, n), knowledge.contiguous().view(-1, n)
    rows = torch.empty(x2.shape[0], device=x.device, dtype=torch.float32)
    L.check(L.lib().cti_kd_rows_fwd(x2.data_ptr(), k2.data_ptr(), rows.data_ptr(), x2.shape[0], n, float(T), _stream()), "cti_kd_rows_fwd")
    return rows


def kd_rows_bwd(x, knowledge, upstream, coef, T, dx=None, beta=0.0):
    n = x.shape[-1]
    xc, kc = x.contiguous(), knowledge.contiguous()
    if dx is None:
        dx = torch.empty_like(xc)
        beta = 0.0
    L.check(L.lib().cti_kd_rows_bwd(xc.data_ptr(), kc.data_ptr(), _ptr(upstream), float(coef), dx.data_ptr(), xc.numel() // n, n, float(T),
                                    float(beta), _stream()), "cti_kd_rows_bwd")
    return dx
