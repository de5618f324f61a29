"""Context = one anet_ctx (device + stream + scratch).  One per host thread per device."""
import ctypes
from . import _lib

_contexts = {}


class Context:
    def __init__(self, device=0):
        lib = _lib.load()
        h = ctypes.c_void_p()
        rc = lib.anet_create(int(device), ctypes.byref(h))
        if rc != _lib.ANET_OK:
            msg = lib.anet_last_error(None)
            raise _lib.AnetError(rc, msg.decode() if msg else "?")
        self.lib = lib
        self.handle = h
        self.device = int(device)

    def check(self, rc):
        _lib.check(self.handle, rc)

    @property
    def compute_units(self):
        """anet_compute_units: the device's compute units, which every launch-shape threshold of the library scales with."""
        return int(self.lib.anet_compute_units(self.handle))

    def synchronize(self):
        self.check(self.lib.anet_synchronize(self.handle))

    def set_cancel_flag(self, flag):
        """anet_set_cancel_flag: lbfgs_optimize's progress callback (lbfgs.hpp:580-587) as a cancel word.  `flag`: an int32
        torch CUDA tensor of one element (kept alive here), a raw device-visible address, or None to clear.  While the word
        is non-zero every problem of the one-launch MINCO L-BFGS calls on this context stops after the iteration it is in
        with status LBFGS_CANCELED."""
        addr = None
        if flag is not None:
            addr = int(flag.data_ptr()) if hasattr(flag, "data_ptr") else int(flag)
        self._cancel_keepalive = flag
        self.check(self.lib.anet_set_cancel_flag(self.handle, ctypes.c_void_p(addr) if addr else None))

    def close(self):
        if getattr(self, "handle", None):
            self.lib.anet_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def default_context(device=0):
    ctx = _contexts.get(device)
    if ctx is None:
        ctx = _contexts[device] = Context(device)
    return ctx
