"""Per-forward context: everything a launch needs to know that is not an argument of the op.

`RegTR.forward` opens one (`with context.forward(device, ...) as ctx`) and every op called underneath reads it through
`context.current()`.  Contexts live on a THREAD-LOCAL stack, so two models driven from two host threads (each on its own HIP stream
or its own GPU) never see each other's device pin, operand format, status word or timing lists -- the state that used to sit in
module globals of ops.py / _lib.py.  Outside any forward (tests calling ops directly) a neutral default context applies: launch
device = torch's current device, bf16x3 operands, no status word, no recording.
"""
import threading

import torch

STATUS_F16_RANGE = 1          # include/regtr_hip.h REGTR_STATUS_*
STATUS_NONFINITE_POSE = 2


class ForwardContext:
    __slots__ = ('device_index', 'f16_pair', 'force_x3', 'status', 'gather_records', 'mha_records', 'gemm_records', 'f16_range_log',
                 'reference_order', '_dev_ctx', '_prev')

    def __init__(self, device=None, f16_pair=False, force_x3=False, status=None, gather_records=None, mha_records=None,
                 gemm_records=None, f16_range_log=None, reference_order=False):
        if device is None:
            self.device_index = None              # ask torch at launch time
        else:
            device = torch.device(device)
            if device.type != 'cuda':
                raise RuntimeError('regtr_amd ops need GPU tensors (no CPU fallback)')
            self.device_index = device.index if device.index is not None else torch.cuda.current_device()
        self.f16_pair = bool(f16_pair)            # float32-grade contractions in the f16 pair format where the kernels serve the shape
        self.force_x3 = bool(force_x3)            # the range fallback: six-term bf16 split everywhere (float32's operand range)
        self.status = status                      # int32[1] device tensor the kernels OR REGTR_STATUS_* bits into, or None
        self.gather_records = gather_records      # lists bench.py hands in to time launches with HIP events on the launch stream
        self.mha_records = mha_records
        self.gemm_records = gemm_records
        self.f16_range_log = f16_range_log        # audits: (M, N, K, max |A|, max |W|) of every f16 pair launch (synchronises)
        self.reference_order = bool(reference_order)    # cpp_wrappers: the reference CPU ops' own row / tie orders (parity mode)
        self._dev_ctx = None
        self._prev = None

    def derive(self, **kw):
        """A context like this one with some fields replaced (same device)."""
        c = ForwardContext.__new__(ForwardContext)
        for k in ForwardContext.__slots__:
            setattr(c, k, getattr(self, k))
        c._dev_ctx = c._prev = None
        for k, v in kw.items():
            setattr(c, k, v)
        return c

    def status_ptr(self):
        return self.status.data_ptr() if self.status is not None else None

    def __enter__(self):
        # kernels go to torch's current stream OF THE CURRENT DEVICE: make the context's device current for the enclosed launches
        if self.device_index is not None:
            self._dev_ctx = torch.cuda.device(self.device_index)
            self._dev_ctx.__enter__()
        self._prev = getattr(_tls, 'ctx', None)
        _tls.ctx = self
        return self

    def __exit__(self, *exc):
        _tls.ctx = self._prev
        self._prev = None
        if self._dev_ctx is not None:
            ctx, self._dev_ctx = self._dev_ctx, None
            return ctx.__exit__(*exc)
        return False


_tls = threading.local()
_DEFAULT = ForwardContext()


def current():
    c = getattr(_tls, 'ctx', None)
    return c if c is not None else _DEFAULT


def set_thread_default(**kw):
    """Replaces fields of THIS thread's base context (what applies outside any `with` block) and returns their previous values: the setter
    behind cpp_wrappers.set_reference_order.  The process-wide default context is never mutated: a thread that has none gets its own copy."""
    c = getattr(_tls, 'ctx', None)
    if c is None:
        c = _tls.ctx = _DEFAULT.derive()
    prev = {k: getattr(c, k) for k in kw}
    for k, v in kw.items():
        setattr(c, k, v)
    return prev


def forward(device, **kw):
    """`with context.forward(dev, f16_pair=..., status=...) as ctx:` -- a fresh context on this thread's stack.  Recording lists and the
    audit log of an enclosing context (bench.py / tests wrap model calls in `context.recording(...)`) are inherited."""
    outer = current()
    for k in ('gather_records', 'mha_records', 'gemm_records', 'f16_range_log', 'reference_order'):
        if k not in kw:
            kw[k] = getattr(outer, k)
    return ForwardContext(device, **kw)


def recording(**kw):
    """`with context.recording(gather_records=[...]):` -- asks the forwards run inside (on this thread) to time their launches /
    log their operand ranges into the given lists."""
    return current().derive(**kw)


def on_device(device):
    """`with on_device(dev):` -- pins the launch device for ops called directly (tests, cpp_wrappers), keeping the rest of the
    enclosing context."""
    device = torch.device(device)
    if device.type != 'cuda':
        raise RuntimeError('regtr_amd ops need GPU tensors (no CPU fallback)')
    return current().derive(device_index=device.index if device.index is not None else torch.cuda.current_device())
