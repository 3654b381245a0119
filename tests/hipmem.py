"""Device buffers for the -m gpu tests without torch: hipMalloc / hipMemcpy through ctypes, on the HIP runtime
libjppgpu.so itself is linked against (loading torch's bundled runtime after ours has opened the device fails with
"No HIP GPUs are available"; bench.py, where torch comes first, is not affected)."""
import ctypes as C

import numpy as np

_hip = None


def hip():
    global _hip
    if _hip is None:
        lib = C.CDLL('/opt/rocm/lib/libamdhip64.so.7')
        lib.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        lib.hipFree.argtypes = [C.c_void_p]
        lib.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        lib.hipDeviceSynchronize.argtypes = []
        _hip = lib
    return _hip


class DeviceArray:
    """a device allocation holding `count` elements of numpy dtype `dtype`"""

    def __init__(self, count, dtype):
        self.dtype = np.dtype(dtype)
        self.count = int(count)
        self.nbytes = max(1, self.count * self.dtype.itemsize)
        p = C.c_void_p()
        rc = hip().hipMalloc(C.byref(p), self.nbytes)
        if rc != 0 or not p.value:
            raise RuntimeError('hipMalloc(%d) failed: %d' % (self.nbytes, rc))
        self.ptr = p.value

    @classmethod
    def from_numpy(cls, a):
        a = np.ascontiguousarray(a)
        d = cls(a.size, a.dtype)
        if a.size:
            rc = hip().hipMemcpy(d.ptr, a.ctypes.data, a.nbytes, 1)   # hipMemcpyHostToDevice
            if rc != 0:
                raise RuntimeError('hipMemcpy H2D failed: %d' % rc)
        return d

    def to_numpy(self, count=None):
        n = self.count if count is None else int(count)
        out = np.empty(n, dtype=self.dtype)
        if n:
            hip().hipDeviceSynchronize()
            rc = hip().hipMemcpy(out.ctypes.data, self.ptr, n * self.dtype.itemsize, 2)   # hipMemcpyDeviceToHost
            if rc != 0:
                raise RuntimeError('hipMemcpy D2H failed: %d' % rc)
        return out

    def free(self):
        if self.ptr:
            hip().hipFree(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass
