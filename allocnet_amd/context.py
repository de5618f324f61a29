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

    def synchronize(self):
        self.check(self.lib.anet_synchronize(self.handle))

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
