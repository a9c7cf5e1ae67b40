"""Host-side runtime glue: execution context + device-array helpers (torch owns the HBM allocations).

Mirrors what the reference's LinearAlgebraFactory + ExecSpace do for the `HIP` mem-space
(src/LinAlg/LinAlgFactory.cpp:100-180, src/ExecBackends/ExecSpace.hpp:345-457): arrays live in device
memory and only raw pointers cross into the C ABI."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from ._lib import lib, check


import weakref

_LIVE_CONTEXTS = weakref.WeakValueDictionary()   # id(Context) -> Context: a context nobody refers to any more is collected
_KEEP = []             # [tensor, {ids of the contexts whose stream may still use it}]
_KEEP_MAX = 8192


def _release_for(ctx_id):
    """`ctx_id` has synchronised (or is gone): drop it from every entry; entries nobody waits for are freed."""
    keep = []
    for e in _KEEP:
        e[1].discard(ctx_id)
        if e[1]:
            keep.append(e)
    _KEEP[:] = keep


def dptr(t, ctx=None) -> C.c_void_p:
    """Raw device (or host) address of a torch tensor / numpy array / None.

    The C ABI is asynchronous on the context's own stream, which torch's caching allocator knows nothing about: a device
    temporary released right after the call would be recycled for the next allocation while the call's kernels are still
    queued.  Every device tensor whose address crosses the ABI is therefore kept alive until the context that uses it
    (`ctx`; every live context when the caller does not say) has synchronised (`Context.sync`).  The list is bounded: when it
    overflows, the contexts that entries still wait for are synchronised — an idle or forgotten context cannot pin memory."""
    if t is None:
        return C.c_void_p(0)
    if isinstance(t, torch.Tensor):
        assert t.is_contiguous()
        if t.is_cuda:
            # producer side of the same problem: the tensor may have been written by kernels still queued on torch's current
            # stream — the context's stream waits for them (an event, no host synchronisation)
            cur = torch.cuda.current_stream()
            for c in ([ctx] if ctx is not None else list(_LIVE_CONTEXTS.values())):
                if getattr(c, "torch_stream", None) is not None and cur.cuda_stream != c.torch_stream.cuda_stream:
                    c.torch_stream.wait_stream(cur)
            ids = {id(ctx)} if ctx is not None else set(_LIVE_CONTEXTS.keys())
            if ids:
                _KEEP.append([t, ids])
            if len(_KEEP) > _KEEP_MAX:
                for c in list(_LIVE_CONTEXTS.values()):
                    c.sync()
                _KEEP.clear()
        return C.c_void_p(t.data_ptr())
    if isinstance(t, np.ndarray):
        assert t.flags["C_CONTIGUOUS"]
        return C.c_void_p(t.ctypes.data)
    raise TypeError(type(t))


def dev(a, dtype=None, device="cuda"):
    """numpy -> device tensor (fp64 / int32)."""
    t = torch.as_tensor(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.to(device).contiguous()


class Context:
    """Owns a hiopamd_ctx (HIP stream + reduction scratch).  All kernels of one context are stream-ordered."""

    def __init__(self, device_index: int = 0):
        if not torch.cuda.is_available():
            raise RuntimeError("hiop_amd needs a MI355X (gfx950) device: no HIP device is visible and there is no CPU path")
        torch.cuda.set_device(device_index)
        self._L = lib()
        h = C.c_void_p()
        check(self._L.hiopamd_ctx_create(C.byref(h), None), "hiopamd_ctx_create")
        self.h = h
        self.stream_ptr = self._L.hiopamd_ctx_stream(self.h)
        self.torch_stream = torch.cuda.ExternalStream(self.stream_ptr)
        self._children = []   # weakrefs of objects holding C handles that reference this context
        _LIVE_CONTEXTS[id(self)] = self

    def _register(self, obj):
        self._children.append(weakref.ref(obj))

    def sync(self):
        check(self._L.hiopamd_ctx_sync(self.h), "hiopamd_ctx_sync")
        _release_for(id(self))

    def close(self):
        if self.h is not None:
            # destroy dependants first: their C structs keep a pointer to this context
            for w in self._children:
                o = w()
                if o is not None:
                    o.close()
            self._children = []
            _LIVE_CONTEXTS.pop(id(self), None)
            self._L.hiopamd_ctx_sync(self.h)
            _release_for(id(self))
            self._L.hiopamd_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- small call helpers ----
    def call(self, name, *args):
        fn = getattr(self._L, name)
        conv = []
        for a in args:
            if isinstance(a, (torch.Tensor, np.ndarray)) or a is None:
                conv.append(dptr(a, self))
            else:
                conv.append(a)
        check(fn(self.h, *conv), name)

    def reduce_double(self, name, *args) -> float:
        out = C.c_double(0.0)
        self.call(name, *args, C.byref(out))
        return out.value

    def reduce_int(self, name, *args) -> int:
        out = C.c_int(0)
        self.call(name, *args, C.byref(out))
        return out.value

    def reduce_int64(self, name, *args) -> int:
        out = C.c_int64(0)
        self.call(name, *args, C.byref(out))
        return out.value

    def init_rccl_from_torch_distributed(self):
        """One RCCL communicator per context; the 128-byte unique id travels over torch.distributed."""
        import torch.distributed as dist
        rank, size = dist.get_rank(), dist.get_world_size()
        uid = (C.c_ubyte * 128)()
        if rank == 0:
            check(self._L.hiopamd_rccl_unique_id(uid), "hiopamd_rccl_unique_id")
        t = torch.tensor(list(uid), dtype=torch.uint8, device="cuda")
        dist.broadcast(t, 0)
        arr = (C.c_ubyte * 128)(*t.cpu().tolist())
        check(self._L.hiopamd_ctx_init_rccl(self.h, arr, rank, size), "hiopamd_ctx_init_rccl")
