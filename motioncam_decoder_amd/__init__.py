"""motioncam_decoder_amd -- MI355X (gfx950) MCRAW frame-decode path.

Python is plumbing only: this module loads the C-ABI shared library
(``lib/libmcraw_hip.so``, declared in ``include/mcraw_hip.h``) with ctypes and
hands it device pointers (e.g. ``torch.Tensor.data_ptr()``).  The product is
the HIP library and the C++ ``motioncam::Decoder`` facade under ``host/``.

There is no CPU decode path here: if the HIP library is missing or no GPU is
present, every decode call raises.
"""
import ctypes as C
import os

__all__ = ["lib_path", "load", "Context", "Pool", "Frame", "McrawError", "TYPE_LEGACY", "TYPE_BLOCK",
           "MEM_DEVICE", "MEM_HOST", "KERNELS", "ABI_SYMBOLS"]

_PKG = os.path.dirname(os.path.abspath(__file__))

TYPE_LEGACY = 6
TYPE_BLOCK = 7
MEM_DEVICE = 0
MEM_HOST = 1

# status bits (include/mcraw_hip.h)
E_ARGS, E_HEADER, E_TRUNCATED, E_SIDESTREAM, E_CAPACITY, E_DEVICE = 0x1, 0x2, 0x4, 0x8, 0x10, 0x100

KERNELS = {"k7_side": 0, "k7_tiles": 3, "k6_decode": 6}

# every symbol include/mcraw_hip.h declares
ABI_SYMBOLS = [
    "mcraw_ctx_create", "mcraw_ctx_destroy", "mcraw_last_error", "mcraw_decode7", "mcraw_decode6",
    "mcraw_decode_batch", "mcraw_ctx_synchronize", "mcraw_ctx_profile", "mcraw_ctx_kernel_ms",
    "mcraw_host_alloc", "mcraw_host_free", "mcraw_ctx_set_post", "mcraw_decode_batch_async", "mcraw_ticket_wait",
    "mcraw_ctx_profile_every", "mcraw_legacy_launch_order", "mcraw_shard_of", "mcraw_shard_count", "mcraw_pool_create", "mcraw_pool_destroy", "mcraw_pool_last_error",
    "mcraw_pool_size", "mcraw_pool_device", "mcraw_pool_numa_cpus", "mcraw_pool_ctx", "mcraw_pool_set_post",
    "mcraw_pool_host_alloc", "mcraw_pool_decode_batch", "mcraw_pool_decode_batch_async", "mcraw_pool_ticket_wait",
    "mcraw_pool_decode_batch_device", "mcraw_ctx_xcd_runs", "mcraw_pool_synchronize", "mcraw_tile_order",
    "mcraw_ctx_last_serial", "mcraw_ctx_batch_status", "mcraw_ctx_errors", "mcraw_ctx_side_parts", "mcraw_ctx_host_way",
]

POST_BLACK, POST_PACK12, POST_PACK10, POST_PACK14 = 1, 2, 4, 8
_PACK_FLAG = {16: 0, 12: POST_PACK12, 10: POST_PACK10, 14: POST_PACK14}


class Post(C.Structure):
    """struct mcraw_post (include/mcraw_hip.h)."""
    _fields_ = [("flags", C.c_uint32), ("black", C.c_uint16 * 4)]


class McrawError(RuntimeError):
    pass


class Frame(C.Structure):
    """struct mcraw_frame (include/mcraw_hip.h)."""
    _fields_ = [("in_", C.c_void_p), ("len", C.c_size_t), ("width", C.c_int32), ("height", C.c_int32),
                ("type", C.c_int32), ("reserved", C.c_int32), ("out", C.c_void_p), ("out_capacity", C.c_size_t)]


def lib_path():
    # MCRAW_LIB_PATH: load another build of the same ABI (A/B timing of kernel variants on one box)
    return os.environ.get("MCRAW_LIB_PATH") or os.path.join(_PKG, "lib", "libmcraw_hip.so")


_lib = None


def load():
    """Load libmcraw_hip.so; raises McrawError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    p = lib_path()
    if not os.path.exists(p) and "MCRAW_LIB_PATH" not in os.environ:
        try:  # a fresh checkout: compile the kernels now (hipcc, ~40 s); this is a build step, not a fallback
            from . import build as _build
            _build.build_hip()
        except Exception as e:
            raise McrawError("HIP decode library not built and building it failed (%s): %s; "
                             "there is no CPU fallback" % (e, p))
    if not os.path.exists(p):
        raise McrawError("HIP decode library not built: %s (run `python -m motioncam_decoder_amd.build`); "
                         "there is no CPU fallback" % p)
    # A process that also uses torch must share ONE HIP runtime with it: torch bundles its
    # own libamdhip64 (same SONAME as /opt/rocm's).  Loading torch first makes this library
    # bind to the runtime torch initialises; the other order maps two runtimes and the
    # second one sees no device.  (A C++ host without torch simply uses /opt/rocm's.)
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(p)
    lib.mcraw_ctx_create.restype = C.c_int
    lib.mcraw_ctx_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
    lib.mcraw_ctx_destroy.restype = None
    lib.mcraw_ctx_destroy.argtypes = [C.c_void_p]
    lib.mcraw_last_error.restype = C.c_char_p
    lib.mcraw_last_error.argtypes = []
    for name in ("mcraw_decode7", "mcraw_decode6"):
        fn = getattr(lib, name)
        fn.restype = C.c_size_t
        fn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t]
    lib.mcraw_decode_batch.restype = C.c_int
    lib.mcraw_decode_batch.argtypes = [C.c_void_p, C.POINTER(Frame), C.c_int, C.c_int, C.c_void_p,
                                       C.POINTER(C.c_size_t), C.POINTER(C.c_int32)]
    lib.mcraw_ctx_synchronize.restype = C.c_int
    lib.mcraw_ctx_synchronize.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.c_int]
    lib.mcraw_ctx_last_serial.restype = C.c_uint64
    lib.mcraw_ctx_last_serial.argtypes = [C.c_void_p]
    lib.mcraw_ctx_batch_status.restype = C.c_int
    lib.mcraw_ctx_batch_status.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.c_int32), C.c_int]
    lib.mcraw_ctx_errors.restype = C.c_int32
    lib.mcraw_ctx_errors.argtypes = [C.c_void_p, C.c_int]
    lib.mcraw_ctx_profile.restype = C.c_int
    lib.mcraw_ctx_profile.argtypes = [C.c_void_p, C.c_int]
    lib.mcraw_ctx_profile_every.restype = C.c_int
    lib.mcraw_ctx_profile_every.argtypes = [C.c_void_p, C.c_int]
    lib.mcraw_ctx_kernel_ms.restype = C.c_int
    lib.mcraw_ctx_kernel_ms.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int), C.c_int]
    lib.mcraw_host_alloc.restype = C.c_void_p
    lib.mcraw_host_alloc.argtypes = [C.c_size_t]
    lib.mcraw_host_free.restype = None
    lib.mcraw_host_free.argtypes = [C.c_void_p]
    lib.mcraw_decode_batch_async.restype = C.c_int
    lib.mcraw_decode_batch_async.argtypes = [C.c_void_p, C.POINTER(Frame), C.c_int, C.POINTER(C.c_void_p)]
    lib.mcraw_ticket_wait.restype = C.c_int
    lib.mcraw_ticket_wait.argtypes = [C.c_void_p, C.POINTER(C.c_size_t), C.POINTER(C.c_int32)]
    lib.mcraw_ctx_set_post.restype = C.c_int
    lib.mcraw_ctx_set_post.argtypes = [C.c_void_p, C.POINTER(Post)]
    lib.mcraw_shard_of.restype = C.c_int
    lib.mcraw_shard_of.argtypes = [C.c_long, C.c_int]
    lib.mcraw_shard_count.restype = C.c_int
    lib.mcraw_shard_count.argtypes = [C.c_long, C.c_int, C.c_int]
    lib.mcraw_pool_create.restype = C.c_int
    lib.mcraw_pool_create.argtypes = [C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_void_p)]
    lib.mcraw_pool_destroy.restype = None
    lib.mcraw_pool_destroy.argtypes = [C.c_void_p]
    lib.mcraw_pool_last_error.restype = C.c_char_p
    lib.mcraw_pool_last_error.argtypes = []
    for name in ("mcraw_pool_size",):
        getattr(lib, name).restype = C.c_int
        getattr(lib, name).argtypes = [C.c_void_p]
    for name in ("mcraw_pool_device", "mcraw_pool_numa_cpus"):
        getattr(lib, name).restype = C.c_int
        getattr(lib, name).argtypes = [C.c_void_p, C.c_int]
    lib.mcraw_pool_ctx.restype = C.c_void_p
    lib.mcraw_pool_ctx.argtypes = [C.c_void_p, C.c_int]
    lib.mcraw_pool_set_post.restype = C.c_int
    lib.mcraw_pool_set_post.argtypes = [C.c_void_p, C.POINTER(Post)]
    lib.mcraw_pool_host_alloc.restype = C.c_void_p
    lib.mcraw_pool_host_alloc.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
    lib.mcraw_pool_decode_batch.restype = C.c_int
    lib.mcraw_pool_decode_batch.argtypes = [C.c_void_p, C.POINTER(Frame), C.c_int, C.POINTER(C.c_size_t), C.POINTER(C.c_int32)]
    lib.mcraw_tile_order.restype = C.c_uint32
    lib.mcraw_tile_order.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32]
    lib.mcraw_pool_synchronize.restype = C.c_int
    lib.mcraw_pool_synchronize.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.c_int]
    lib.mcraw_ctx_xcd_runs.restype = C.c_int
    lib.mcraw_ctx_xcd_runs.argtypes = [C.c_void_p]
    lib.mcraw_ctx_side_parts.restype = C.c_int
    lib.mcraw_ctx_side_parts.argtypes = [C.c_void_p]
    lib.mcraw_ctx_host_way.restype = C.c_int
    lib.mcraw_ctx_host_way.argtypes = [C.c_void_p]
    lib.mcraw_pool_decode_batch_device.restype = C.c_int
    lib.mcraw_pool_decode_batch_device.argtypes = [C.c_void_p, C.POINTER(Frame), C.c_int, C.POINTER(C.c_size_t), C.POINTER(C.c_int32)]
    lib.mcraw_pool_decode_batch_async.restype = C.c_int
    lib.mcraw_pool_decode_batch_async.argtypes = [C.c_void_p, C.POINTER(Frame), C.c_int, C.POINTER(C.c_void_p)]
    lib.mcraw_pool_ticket_wait.restype = C.c_int
    lib.mcraw_pool_ticket_wait.argtypes = [C.c_void_p, C.POINTER(C.c_size_t), C.POINTER(C.c_int32)]
    _lib = lib
    return lib


class Pool:
    """Owner of one ``mcraw_pool``: several GPUs of one node, frame i -> member i mod size."""

    def __init__(self, devices=None):
        self._lib = load()
        h = C.c_void_p()
        devices = list(devices or [])
        arr = (C.c_int * max(len(devices), 1))(*devices)
        rc = self._lib.mcraw_pool_create(arr if devices else None, len(devices), C.byref(h))
        if rc != 0 or not h.value:
            raise McrawError("mcraw_pool_create failed (%d): %s" % (rc, self._lib.mcraw_pool_last_error().decode()))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self._lib.mcraw_pool_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def size(self):
        return self._lib.mcraw_pool_size(self._h)

    def devices(self):
        return [self._lib.mcraw_pool_device(self._h, m) for m in range(self.size)]

    def numa_cpus(self):
        return [self._lib.mcraw_pool_numa_cpus(self._h, m) for m in range(self.size)]

    def host_alloc(self, member, nbytes):
        return self._lib.mcraw_pool_host_alloc(self._h, member, nbytes)

    def set_post(self, black=None, pack12=False, bits=None):
        nb = int(bits) if bits else (12 if pack12 else 16)
        if black is None and nb == 16:
            rc = self._lib.mcraw_pool_set_post(self._h, None)
        else:
            p = Post()
            if black is not None:
                p.flags |= POST_BLACK
                for i in range(4):
                    p.black[i] = int(black[i])
            p.flags |= _PACK_FLAG[nb]
            rc = self._lib.mcraw_pool_set_post(self._h, C.byref(p))
        if rc != 0:  # (a rejected stage must not decode plain mosaics silently)
            raise McrawError("mcraw_pool_set_post failed (%d): %s" % (rc, self._lib.mcraw_last_error().decode()))

    def decode_batch_device(self, frames, want_status=True):
        """frames: ctypes array from Context.make_frames whose in / out pointers live in the HBM of the GPU that decodes
        the frame: frame i on ``devices()[i % size]``.  Returns (written, status); with want_status=False the members
        only queue their shares (synchronize() waits and returns the statuses)."""
        n = len(frames)
        written = (C.c_size_t * max(n, 1))()
        status = (C.c_int32 * max(n, 1))()
        rc = self._lib.mcraw_pool_decode_batch_device(self._h, frames, n, written if want_status else None, status if want_status else None)
        if rc != 0:
            raise McrawError("mcraw_pool_decode_batch_device failed (%d): %s" % (rc, self._lib.mcraw_pool_last_error().decode()))
        return (list(written)[:n], list(status)[:n]) if want_status else None

    def synchronize(self, n):
        """Waits for everything queued; returns the statuses of the calling thread's last queued resident batch.
        ``self.errors``: OR of the statuses of all frames of all queued batches whose outcome became known with this call."""
        status = (C.c_int32 * max(n, 1))()
        rc = self._lib.mcraw_pool_synchronize(self._h, status, n)
        if rc < 0:
            raise McrawError("mcraw_pool_synchronize failed (%d): %s" % (rc, self._lib.mcraw_pool_last_error().decode()))
        self.errors = rc
        return list(status)[:n]

    def decode_batch(self, frames):
        """frames: ctypes array from Context.make_frames (host pointers).  Returns (written, status)."""
        n = len(frames)
        written = (C.c_size_t * max(n, 1))()
        status = (C.c_int32 * max(n, 1))()
        rc = self._lib.mcraw_pool_decode_batch(self._h, frames, n, written, status)
        if rc != 0:
            raise McrawError("mcraw_pool_decode_batch failed (%d): %s" % (rc, self._lib.mcraw_pool_last_error().decode()))
        return list(written)[:n], list(status)[:n]

    def decode_batch_async(self, frames):
        t = C.c_void_p()
        rc = self._lib.mcraw_pool_decode_batch_async(self._h, frames, len(frames), C.byref(t))
        if rc != 0:
            raise McrawError("mcraw_pool_decode_batch_async failed (%d): %s" % (rc, self._lib.mcraw_pool_last_error().decode()))
        return (t, len(frames))

    def wait(self, ticket):
        t, n = ticket
        written = (C.c_size_t * max(n, 1))()
        status = (C.c_int32 * max(n, 1))()
        rc = self._lib.mcraw_pool_ticket_wait(t, written, status)
        if rc != 0:
            raise McrawError("mcraw_pool_ticket_wait failed (%d): %s" % (rc, self._lib.mcraw_pool_last_error().decode()))
        return list(written)[:n], list(status)[:n]


class Context:
    """Owner of one ``mcraw_ctx`` (one HIP device)."""

    def __init__(self, device=-1):
        self._lib = load()
        h = C.c_void_p()
        rc = self._lib.mcraw_ctx_create(device, C.byref(h))
        if rc != 0 or not h.value:
            raise McrawError("mcraw_ctx_create failed (%d): %s" % (rc, self._lib.mcraw_last_error().decode()))
        self._h = h

    def xcd_runs(self):
        """The XCD mapping the library chose for the current large resident batches (mcraw_ctx_xcd_runs)."""
        return self._lib.mcraw_ctx_xcd_runs(self._h)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.mcraw_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def make_frames(descs):
        """descs: iterable of (in_ptr, len, width, height, type, out_ptr, out_capacity)."""
        descs = list(descs)
        arr = (Frame * len(descs))()
        for i, (inp, ln, w, h, t, outp, cap) in enumerate(descs):
            arr[i] = Frame(inp, ln, w, h, t, 0, outp, cap)
        return arr

    def decode_batch(self, frames, mem=MEM_DEVICE, stream=None, want_status=True):
        """frames: ctypes array from make_frames.  Returns (written, status) lists when
        want_status (synchronises), else None (asynchronous on `stream`)."""
        n = len(frames)
        if want_status:
            written = (C.c_size_t * n)()
            status = (C.c_int32 * n)()
            rc = self._lib.mcraw_decode_batch(self._h, frames, n, mem, stream, written, status)
        else:
            rc = self._lib.mcraw_decode_batch(self._h, frames, n, mem, stream, None, None)
        if rc != 0:
            raise McrawError("mcraw_decode_batch failed (%d): %s" % (rc, self._lib.mcraw_last_error().decode()))
        if want_status:
            return list(written), list(status)
        return None

    def decode_batch_async(self, frames):
        """Host-memory batch, queued: returns a ticket for wait()."""
        t = C.c_void_p()
        rc = self._lib.mcraw_decode_batch_async(self._h, frames, len(frames), C.byref(t))
        if rc != 0:
            raise McrawError("mcraw_decode_batch_async failed (%d): %s" % (rc, self._lib.mcraw_last_error().decode()))
        return (t, len(frames))

    def wait(self, ticket):
        """Blocks for that batch; returns (written, status)."""
        t, n = ticket
        written = (C.c_size_t * max(n, 1))()
        status = (C.c_int32 * max(n, 1))()
        rc = self._lib.mcraw_ticket_wait(t, written, status)
        if rc != 0:
            raise McrawError("mcraw_ticket_wait failed (%d): %s" % (rc, self._lib.mcraw_last_error().decode()))
        return list(written)[:n], list(status)[:n]

    def synchronize(self, nframes=0):
        status = (C.c_int32 * max(nframes, 1))()
        rc = self._lib.mcraw_ctx_synchronize(self._h, status if nframes else None, nframes)
        if rc != 0:
            raise McrawError("mcraw_ctx_synchronize failed (%d): %s" % (rc, self._lib.mcraw_last_error().decode()))
        return list(status)[:nframes]

    def side_parts(self):
        """(parts of the bits stream, parts of the refs stream) k7_side was measured to run fastest with, or None."""
        v = int(self._lib.mcraw_ctx_side_parts(self._h))
        return None if v < 0 else (v >> 4, v & 15)

    def host_way(self):
        """How the status words of large host-memory batches come home: 0 fetched, 1 sent behind the kernels, None: still comparing."""
        v = int(self._lib.mcraw_ctx_host_way(self._h))
        return None if v < 0 else v

    def last_serial(self):
        return int(self._lib.mcraw_ctx_last_serial(self._h))

    def batch_status(self, serial, nframes):
        """Statuses of the device-memory batch `serial` that was submitted without a status request (None: not one of the last 64)."""
        status = (C.c_int32 * max(nframes, 1))()
        rc = self._lib.mcraw_ctx_batch_status(self._h, serial, status, nframes)
        if rc < 0:
            raise McrawError("mcraw_ctx_batch_status failed (%d): %s" % (rc, self._lib.mcraw_last_error().decode()))
        return None if rc else list(status)[:nframes]

    def errors(self, reset=True):
        return int(self._lib.mcraw_ctx_errors(self._h, 1 if reset else 0))

    def set_post(self, black=None, pack12=False, bits=None):
        """Fused post-decode stage of the batches to come: black levels (4 values, CFA order
        (row & 1) * 2 + (col & 1)) and/or strip rows of `bits` = 10, 12 or 14 bits per sample (pack12=True: 12);
        no arguments = the plain uint16 mosaic."""
        nb = int(bits) if bits else (12 if pack12 else 16)
        if black is None and nb == 16:
            rc = self._lib.mcraw_ctx_set_post(self._h, None)
        else:
            p = Post()
            if black is not None:
                p.flags |= POST_BLACK
                for i in range(4):
                    p.black[i] = int(black[i])
            p.flags |= _PACK_FLAG[nb]
            rc = self._lib.mcraw_ctx_set_post(self._h, C.byref(p))
        if rc != 0:
            raise McrawError("mcraw_ctx_set_post failed (%d): %s" % (rc, self._lib.mcraw_last_error().decode()))

    def profile(self, enable=True, only=None, every=1):
        """Bracket kernel launches with events: all kernels, or just the names in `only`; every `every`-th launch."""
        self._lib.mcraw_ctx_profile_every(self._h, max(1, int(every)))
        if only:
            mode = 0
            for name in only:
                mode |= 2 << KERNELS[name]
        else:
            mode = 1 if enable else 0
        self._lib.mcraw_ctx_profile(self._h, mode)

    def kernel_ms(self, name, reset=False):
        ms = C.c_double()
        n = C.c_int()
        rc = self._lib.mcraw_ctx_kernel_ms(self._h, KERNELS[name], C.byref(ms), C.byref(n), 1 if reset else 0)
        if rc != 0:
            raise McrawError("mcraw_ctx_kernel_ms failed (%d)" % rc)
        return ms.value, n.value
