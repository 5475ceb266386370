"""ctypes binding of libssdr_al.so (C ABI: include/ssdr_al.h).

The library is the gfx950 HIP build made by ``ssdr-al_amd/csrc/Makefile`` (``__graft_entry__.build()``).
There is no CPU fallback: if the shared object is missing, or no HIP device is present, every op raises.
``SSDR_AL_LIBRARY`` may point at another build of the same ABI (the CPU logic-test build used by the
``-m "not gpu"`` tests sets it explicitly); nothing selects such a build implicitly.
"""
import ctypes as C
import os

import numpy as np

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEFAULT_PATH = os.path.join(_PKG, "libssdr_al.so")

SSDR_OK = 0
ORDER_REFERENCE, ORDER_KEY = 0, 1

_lib = None
_forced_path = None
_libs = {}          # path -> its CDLL, kept for the life of the process: DevArray's pool of freed buffers is keyed by the object, and a collected CDLL's id() can
                    # come back as another library's (the tests switch between the CPU logic build and the gfx950 build: a host pointer handed to hipMemcpy)


class SsdrError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__("libssdr_al status %d: %s" % (status, msg))
        self.status = status


def lib_path():
    return _forced_path or os.environ.get("SSDR_AL_LIBRARY", DEFAULT_PATH)


def use(path):
    """Bind to another build of the same C ABI (tests switch between the CPU logic build and the gfx950
    build); ``None`` returns to the default resolution."""
    global _lib, _forced_path
    _forced_path = path
    _lib = None


def lib():
    global _lib
    if _lib is None:
        path = lib_path()
        if path in _libs:
            _lib = _libs[path]
            return _lib
        if not os.path.exists(path):
            raise SsdrError(-1, "%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                                "(hipcc --offload-arch=gfx950); there is no CPU fallback" % path)
        L = C.CDLL(path)
        vp, sz, f32, i32 = C.c_void_p, C.c_size_t, C.c_float, C.c_int
        L.ssdr_version.restype = C.c_char_p
        L.ssdr_last_error.restype = C.c_char_p
        L.ssdr_last_gpu_ms.restype = C.c_float
        L.ssdr_init.argtypes = [i32]
        L.ssdr_stream_sync.argtypes = [vp]
        L.ssdr_stream_create.argtypes = [C.POINTER(vp)]
        L.ssdr_stream_create_priority.argtypes = [C.POINTER(vp), i32]
        L.ssdr_stream_destroy.argtypes = [vp]
        L.ssdr_stream_wait.argtypes = [vp, vp]
        L.ssdr_knn.argtypes = [vp, sz, sz, vp, sz, sz, vp]
        L.ssdr_knn_batch.argtypes = [vp, sz, sz, sz, vp, sz, sz, vp]
        L.ssdr_knn_batch_i32.argtypes = [vp, sz, sz, sz, vp, sz, sz, vp]
        L.ssdr_knn_batch_dev.argtypes = [vp, sz, sz, sz, vp, sz, sz, vp, vp]
        L.ssdr_knn_status.argtypes = [vp, vp]
        L.ssdr_knn_pyramid.argtypes = [vp, sz, sz, sz, vp, sz, vp, vp, vp]
        L.ssdr_knn_pyramid_dev.argtypes = [vp, sz, sz, sz, vp, sz, vp, vp, vp, vp]
        L.ssdr_grid_subsample.argtypes = [vp, sz, vp, sz, vp, sz, f32, i32, C.POINTER(sz)]
        L.ssdr_grid_subsample_fetch.argtypes = [vp, vp, vp]
        L.ssdr_grid_subsample_dev.argtypes = [vp, sz, vp, sz, vp, sz, f32, i32, vp, vp, vp, vp, vp]
        L.ssdr_randla_create.argtypes = [i32, vp, i32, i32, i32, C.POINTER(vp)]
        L.ssdr_randla_num_layers.argtypes = [vp]
        L.ssdr_randla_layer_shape.argtypes = [vp, i32, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]
        L.ssdr_randla_set_layer.argtypes = [vp, i32, vp, vp]
        L.ssdr_randla_set_precision.argtypes = [vp, i32]
        L.ssdr_select_status.argtypes = [vp, C.POINTER(i32)]
        L.ssdr_randla_set_formulation.argtypes = [vp, i32]
        L.ssdr_randla_destroy.argtypes = [vp]
        L.ssdr_randla_destroy.restype = None
        L.ssdr_randla_infer_dev.argtypes = [vp, sz, sz, vp, vp, vp, vp, vp, vp, vp, vp]
        f64 = C.c_double
        L.ssdr_point_uncertainty_dev.argtypes = [vp, sz, i32, i32, vp, vp, vp]
        L.ssdr_region_stats_dev.argtypes = [vp, vp, vp, vp, sz, i32, i32, vp, vp, vp, vp]
        L.ssdr_dominant_label_dev.argtypes = [vp, vp, vp, sz, i32, vp, vp, vp]
        L.ssdr_clsbal_dev.argtypes = [vp, sz, vp, vp, sz, vp, vp]
        L.ssdr_class_hist_dev.argtypes = [vp, sz, vp, vp, sz, vp, vp]
        L.ssdr_clsbal_hist_dev.argtypes = [vp, sz, vp, sz, vp, vp]
        L.ssdr_rank_regions_dev.argtypes = [vp, sz, vp, vp]
        L.ssdr_segment_mean_features_dev.argtypes = [vp, i32, vp, vp, vp, vp, vp, sz, vp, vp]
        L.ssdr_widen_f32_f64_dev.argtypes = [vp, sz, vp, vp, vp]
        L.ssdr_knn_batch_distance_pick.argtypes = [vp, sz, sz, sz, vp, sz, sz, vp, C.c_uint32]
        L.ssdr_knn_graph_dev.argtypes = [vp, sz, sz, sz, vp, vp, vp, vp, vp]
        L.ssdr_geof_dev.argtypes = [vp, sz, vp, sz, vp, vp]
        L.ssdr_cloud_graph_dev.argtypes = [vp, vp, vp, vp, sz, sz, i32, vp, vp, vp, vp]
        L.ssdr_propagate_dev.argtypes = [vp, sz, vp, vp, i32, vp, vp, vp]
        L.ssdr_cloud_graph_batch_dev.argtypes = [vp, vp, vp, vp, vp, vp, sz, sz, sz, i32, vp, vp, vp, vp]
        L.ssdr_propagate_batch_dev.argtypes = [vp, vp, vp, sz, sz, vp, vp, i32, vp, vp, vp]
        L.ssdr_fps_dev.argtypes = [vp, sz, i32, i32, sz, vp, vp]
        L.ssdr_create_adj_dev.argtypes = [vp, sz, i32, vp, vp, vp, vp, sz, sz, vp, vp, vp, vp]
        L.ssdr_gcn_fps_sampling_dev.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, sz, vp, vp, sz, vp, vp, sz, sz, i32, i32, i32, i32, sz, sz, sz, sz, sz, vp, vp]
        L.ssdr_fps_superpoint_dev.argtypes = [vp, vp, sz, i32, sz, vp, vp]
        L.ssdr_gcn_fps_sharded_local_dev.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, sz, sz, vp, sz, vp, vp, i32, i32, sz, sz, sz, i32, i32, sz, sz, sz, sz, sz, vp, vp, vp]
        L.ssdr_event_create.argtypes = [C.POINTER(vp)]; L.ssdr_event_record.argtypes = [vp, vp]; L.ssdr_stream_wait_event.argtypes = [vp, vp]; L.ssdr_event_destroy.argtypes = [vp]
        L.ssdr_select_set_chamfer_mode.argtypes = [i32]
        L.ssdr_gcn_fps_sampling_rows.argtypes = [vp, C.POINTER(vp), C.POINTER(sz)]
        L.ssdr_kcenter_gathered_dev.argtypes = [vp, vp, i32, sz, sz, vp, sz, sz, sz, vp, vp, vp, vp]
        L.ssdr_fps_gathered_dev.argtypes = [vp, vp, i32, sz, sz, i32, i32, sz, vp, vp, vp]
        L.ssdr_kcenter_dev.argtypes = [vp, sz, i32, vp, sz, sz, vp, vp]
        L.ssdr_tile_select_dev.argtypes = [vp, vp, i32, vp, sz, vp, sz, vp, vp, f32, vp, vp, vp, vp]
        L.ssdr_grid_subsample_batch_dev.argtypes = [vp, vp, sz, vp, sz, vp, sz, f32, vp, vp, vp, vp, vp]
        L.ssdr_tile_select_batch_dev.argtypes = [vp, vp, i32, vp, vp, sz, vp, sz, vp, vp, f32, vp, vp, vp, vp, vp, vp]
        L.ssdr_tile_select_possibility_dev.argtypes = [vp, vp, i32, vp, sz, vp, sz, vp, vp, f32, vp, vp, vp, vp, vp, vp, vp]
        L.ssdr_split3_dev.argtypes = [vp, sz, sz, sz, sz, vp, vp, vp, sz, C.POINTER(sz), vp]
        L.ssdr_chamfer3d_forward_dev.argtypes = [vp, vp, sz, sz, sz, vp, vp, vp, vp, vp]
        L.ssdr_vote_smooth_dev.argtypes = [vp, vp, vp, sz, i32, f64, vp, vp]
        L.ssdr_confusion_dev.argtypes = [vp, i32, vp, vp, sz, vp, vp, vp, vp]
        L.ssdr_main_stream.argtypes = [C.POINTER(vp)]
        L.ssdr_mask_regions_dev.argtypes = [vp, vp, sz, sz, vp, vp]
        L.ssdr_gather_rows_dev.argtypes = [vp, vp, sz, sz, vp, vp]
        L.ssdr_prune_dev.argtypes = [vp, sz, f32, vp, vp, i32, vp, i32, vp, vp, vp, vp, vp, vp]
        L.ssdr_prune_status.argtypes = [vp, vp]
        L.ssdr_prof_enable.argtypes = [i32]
        L.ssdr_prof_report.restype = C.c_char_p
        L.ssdr_dev_alloc.argtypes = [sz, C.POINTER(vp)]
        L.ssdr_dev_free.argtypes = [vp]
        L.ssdr_memcpy_h2d.argtypes = [vp, vp, sz]
        L.ssdr_memcpy_d2h.argtypes = [vp, vp, sz]
        L.ssdr_memcpy_h2d_on.argtypes = [vp, vp, sz, vp]
        L.ssdr_memcpy_d2h_on.argtypes = [vp, vp, sz, vp]
        L.ssdr_knn_status_poll.argtypes = [vp, vp]
        L.ssdr_grid_subsample_status.argtypes = [vp, vp]
        L.ssdr_grid_subsample_set_method.argtypes = [i32]
        _lib = _libs[path] = L
    return _lib


def check(status):
    if status != SSDR_OK:
        raise SsdrError(status, lib().ssdr_last_error().decode())


def ptr(a):
    """void* of a C-contiguous numpy array (or None)."""
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def last_gpu_ms():
    return float(lib().ssdr_last_gpu_ms())


_pool = {}          # size class -> [device pointers]; avoids hipMalloc/hipFree (both synchronise) in steady state


def _size_class(nbytes):
    c = 256
    while c < nbytes:
        c *= 2
    return c


class DevArray:
    """A device buffer owned through the C ABI (ssdr_dev_alloc / ssdr_memcpy_*): lets the mirror keep tiles
    resident between stages without any framework dependency.  Freed buffers are recycled by size class."""

    def __init__(self, shape, dtype):
        self.shape = tuple(int(s) for s in shape)
        self.dtype = np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape)) * self.dtype.itemsize
        self._cls = _size_class(max(self.nbytes, 1))
        self._owner_lib = lib()
        free = _pool.get((id(self._owner_lib), self._cls))
        if free:
            self.ptr = free.pop()
        else:
            p = C.c_void_p()
            check(lib().ssdr_dev_alloc(self._cls, C.byref(p)))
            self.ptr = p.value

    @classmethod
    def from_host(cls, a, stream=None):
        """stream: order the copy on that stream and wait for it alone (None: the library stream)"""
        a = np.ascontiguousarray(a)
        d = cls(a.shape, a.dtype)
        if stream is None:
            check(lib().ssdr_memcpy_h2d(d.ptr, ptr(a), d.nbytes))
        else:
            check(lib().ssdr_memcpy_h2d_on(d.ptr, ptr(a), d.nbytes, stream))
        return d

    @property
    def __cuda_array_interface__(self):
        """Zero-copy view for frameworks (torch.as_tensor(arr, device="cuda")): the RCCL exchanges run on these buffers."""
        return {"shape": self.shape, "typestr": self.dtype.str, "data": (int(self.ptr), False), "version": 2, "strides": None}

    def host_view(self):
        """CPU logic build only (device memory == host memory there): a NumPy view of the buffer."""
        buf = (C.c_char * self.nbytes).from_address(int(self.ptr))
        return np.frombuffer(buf, self.dtype).reshape(self.shape)

    def to_host(self, stream=None):
        out = np.empty(self.shape, self.dtype)
        if stream is None:
            check(lib().ssdr_memcpy_d2h(ptr(out), self.ptr, self.nbytes))
        else:
            check(lib().ssdr_memcpy_d2h_on(ptr(out), self.ptr, self.nbytes, stream))
        return out

    def __del__(self):
        try:
            if self.ptr:
                free = _pool.setdefault((id(self._owner_lib), self._cls), [])
                if len(free) < 64 and self._cls <= (1 << 26):
                    free.append(self.ptr)
                else:
                    self._owner_lib.ssdr_dev_free(self.ptr)
                self.ptr = None
        except Exception:
            pass


def sync(stream=None):
    check(lib().ssdr_stream_sync(stream))
