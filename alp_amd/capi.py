"""ctypes binding of the C ABI in include/alpgpu.h (libalpgpu.so) — Python-side plumbing only.

The codec runs entirely in hand-written HIP kernels behind the C ABI; this module passes raw device
pointers (from torch tensors, which are used purely as an HBM allocator + stream provider) and sizes.
There is no CPU fallback: importing works anywhere (the .so links only the HIP runtime), but creating a
Context without a gfx950 device raises, and a missing libalpgpu.so raises at import.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np
# torch ships its own libamdhip64.so (same SONAME as /opt/rocm's).  It MUST be loaded first so that
# libalpgpu.so binds to the same HIP runtime instance as the tensors/streams it is handed; two runtimes in
# one process do not see each other's streams (and the second one may not see the device at all).
import torch  # noqa: F401

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ALPGPU_LIB", os.path.join(HERE, "libalpgpu.so"))  # ALPGPU_LIB: A/B builds of the same ABI

VECTOR_SIZE = 1024
ROWGROUP_VECTORS = 100
SCHEME_ALP_RD = 1
SCHEME_ALP = 2

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
        "(hipcc --offload-arch=gfx950).  alp_amd has no CPU fallback.")

lib = C.CDLL(LIB_PATH)

# numpy mirrors of the two HBM record types (include/alpgpu.h)
ROWGROUP_DTYPE = np.dtype([
    ("scheme", np.uint8), ("k", np.uint8), ("combos", np.uint8, (10,)), ("rd_rbw", np.uint8), ("rd_lbw", np.uint8),
    ("rd_dict_size", np.uint8), ("pad", np.uint8), ("rd_dict", np.uint16, (8,)),
], align=False)
VECTOR_DTYPE = np.dtype([
    ("packed_off", np.uint64), ("exc_off", np.uint64), ("base", np.int64), ("bw", np.uint8), ("e", np.uint8),
    ("f", np.uint8), ("lbw", np.uint8), ("exc_cnt", np.uint16), ("scheme", np.uint16),
], align=False)
assert ROWGROUP_DTYPE.itemsize == 32 and VECTOR_DTYPE.itemsize == 32


class CColumn(C.Structure):
    _fields_ = [
        ("n_vectors", C.c_uint64), ("n_rowgroups", C.c_uint64), ("d_rowgroups", C.c_void_p), ("d_vectors", C.c_void_p),
        ("d_packed", C.c_void_p), ("packed_capacity", C.c_uint64), ("d_exc", C.c_void_p), ("exc_capacity", C.c_uint64),
        ("d_totals", C.c_void_p), ("packed_bytes_hint", C.c_uint64), ("exc_bytes_hint", C.c_uint64), ("d_rd_order", C.c_void_p), ("alp_rd_rowgroups_hint", C.c_uint64),
    ]


RD_ORDER_STRIDE = 296  # ALPGPU_RD_ORDER_STRIDE


def _sig(name, restype, *argtypes):
    f = getattr(lib, name)
    f.restype = restype
    f.argtypes = list(argtypes)
    return f


_vp, _u64, _sz, _int = C.c_void_p, C.c_uint64, C.c_size_t, C.c_int
_sig("alpgpu_abi_version", _int)
_sig("alpgpu_last_error", C.c_char_p)
_sig("alpgpu_ctx_create", _int, _int, C.POINTER(_vp))
_sig("alpgpu_ctx_destroy", None, _vp)
_sig("alpgpu_set_stream", _int, _vp, _vp)
_sig("alpgpu_synchronize", _int, _vp)
_sig("alpgpu_set_option", _int, _vp, _int, C.c_int64)
OPT_DECODE_VECTORS_PER_WG, OPT_DECODE_PLAIN_STORES, OPT_ENCODE_TWO_PASS, OPT_DEBUG_FORCE_STALL, OPT_CONSUMER_PIPELINED, OPT_ENCODE_ASYNC_INIT, OPT_ENCODE_KERNEL, OPT_DECODE_PAIRING = 1, 2, 3, 4, 5, 6, 7, 8
OPT_DECODE_PATCH_AFTER, OPT_ENCODE_UNORDERED, OPT_DECODE_RESIDENCY_PAD = 9, 10, 11
OPT_DECODE_READ_AHEAD, OPT_DECODE_READ_AHEAD_US, OPT_DECODE_SEGMENTS, OPT_DECODE_UNHINTED = 12, 13, 14, 15
ENCODE_KERNEL_LEAN, ENCODE_KERNEL_CLASSIC = 0, 1
_sig("alpgpu_device_info", _int, _vp, C.c_char_p, _sz, C.POINTER(_int), C.POINTER(_u64))
_sig("alpgpu_decode_vectors_per_wg", _int, _vp, C.POINTER(CColumn), _int)
_sig("alpgpu_decode_reads_ahead", _int, _vp, C.POINTER(CColumn), _int)
_sig("alpgpu_decode_runs", _int, _vp, C.POINTER(CColumn))
try:  # (round 6; an A/B library of an earlier round selected with ALPGPU_LIB does not have them)
    _sig("alpgpu_decode_runs_f32", _int, _vp, C.POINTER(CColumn))
    _sig("alpgpu_debug_read_ahead_batches", _int, _vp, C.POINTER(_u64))
    _sig("alpgpu_debug_unhinted_plan", _int, _vp, C.POINTER(_u64 * 6))
    _sig("alpgpu_debug_forget_column", _int, _vp, C.POINTER(CColumn))
except AttributeError:
    pass
_sig("alpgpu_debug_traffic_probe", _int, _vp, _vp, _vp, _u64, C.c_uint32)
try:
    _sig("alpgpu_debug_traffic_probe_with_search", _int, _vp, _vp, _vp, _u64, C.c_uint32, C.POINTER(CColumn))
except AttributeError:  # (an A/B library from before round 5, selected with ALPGPU_LIB: a measurement aid it does not have)
    pass
_sig("alpgpu_malloc_host", _int, _vp, C.POINTER(_vp), _sz)
for _t in ("f64", "f32"):
    _sig("alpgpu_compress_host_" + _t, _int, _vp, _vp, _u64, _vp, _u64, C.POINTER(_u64))
    _sig("alpgpu_decompress_host_" + _t, _int, _vp, _vp, _u64, _vp, _u64, C.POINTER(_u64))
for _t in ("f64", "f32"):
    _sig("alpgpu_compress_host_multi_" + _t, _int, C.POINTER(_vp), _int, _vp, _u64, _vp, _u64, C.POINTER(_u64))
    _sig("alpgpu_decompress_host_multi_" + _t, _int, C.POINTER(_vp), _int, _vp, _u64, _vp, _u64, C.POINTER(_u64))
_sig("alpgpu_free_host", _int, _vp, _vp)
_sig("alpgpu_memcpy_h2d_async", _int, _vp, _vp, _vp, _sz)
_sig("alpgpu_debug_decode_probe_f64", _int, _vp, C.POINTER(CColumn), _vp)
_sig("alpgpu_packed_capacity", _u64, _u64)
_sig("alpgpu_exc_capacity", _u64, _u64)
_sig("alpgpu_use_own_stream", _int, _vp)
_sig("alpgpu_decode_f64", _int, _vp, C.POINTER(CColumn), _vp)
_sig("alpgpu_decode_sum_f64", _int, _vp, C.POINTER(CColumn), _vp)
_sig("alpgpu_decode_count_range_f64", _int, _vp, C.POINTER(CColumn), C.c_double, C.c_double, _vp)
_sig("alpgpu_column_sum_f64", _int, _vp, C.POINTER(CColumn), _vp)
_sig("alpgpu_column_sum_f32", _int, _vp, C.POINTER(CColumn), _vp)
_sig("alpgpu_tree_sum_f64", _int, _vp, _vp, _u64, _vp)
_sig("alpgpu_column_validate", _int, _vp, C.POINTER(CColumn), _int, C.POINTER(_u64))
_sig("alpgpu_rowgroup_init_f64", _int, _vp, _vp, _u64, C.POINTER(CColumn))
_sig("alpgpu_encode_vectors_f64", _int, _vp, _vp, _u64, C.POINTER(CColumn))
_sig("alpgpu_encode_f64", _int, _vp, _vp, _u64, C.POINTER(CColumn))
_sig("alpgpu_column_totals", _int, _vp, C.POINTER(CColumn), C.POINTER(_u64), C.POINTER(_u64), C.POINTER(_int))
_sig("alpgpu_pad_tail_f64", _int, _vp, _vp, _u64)
_sig("alpgpu_blob_size", _u64, _u64, _u64, _u64)
_sig("alpgpu_column_to_blob", _int, _vp, C.POINTER(CColumn), _u64, _vp, _u64, C.POINTER(_u64))
_sig("alpgpu_column_from_blob", _int, _vp, _vp, _u64, C.POINTER(CColumn), C.POINTER(_u64))
_sig("alpgpu_ffor_i64", _int, _vp, _vp, _vp, _sz, _vp, _vp, _u64)
_sig("alpgpu_unffor_i64", _int, _vp, _vp, _sz, _vp, _vp, _vp, _u64)
_sig("alpgpu_ffor_u16", _int, _vp, _vp, _vp, _sz, _vp, _vp, _u64)
_sig("alpgpu_unffor_u16", _int, _vp, _vp, _sz, _vp, _vp, _vp, _u64)
_sig("alpgpu_ffor_u8", _int, _vp, _vp, _vp, _sz, _vp, _vp, _u64)
_sig("alpgpu_unffor_u8", _int, _vp, _vp, _sz, _vp, _vp, _vp, _u64)
_sig("alpgpu_falp_f64", _int, _vp, _vp, _sz, _vp, _vp, _vp, _vp, _vp, _u64)
_sig("alpgpu_decode_values_f64", _int, _vp, _vp, _vp, _vp, _vp, _u64)
_sig("alpgpu_patch_f64", _int, _vp, _vp, _vp, _vp, _sz, _vp, _u64)
_sig("alpgpu_encode_simdized_f64", _int, _vp, _vp, _vp, _vp, _sz, _vp, _vp, _vp, _vp, _u64)
_sig("alpgpu_encode_values_f64", _int, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp, _vp, _vp, _vp, _u64)
_sig("alpgpu_analyze_ffor_i64", _int, _vp, _vp, _vp, _vp, _u64)
_sig("alpgpu_encode_value_f64", _int, _vp, _vp, _vp, C.c_uint8, C.c_uint8, _int, _u64)
_sig("alpgpu_encode_value_f32", _int, _vp, _vp, _vp, C.c_uint8, C.c_uint8, _int, _u64)
_sig("alpgpu_rd_encode_vectors_f64", _int, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp, _vp, _vp, _u64)
_sig("alpgpu_rd_decode_vectors_f64", _int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp, _u64)
# single precision (same argument shapes; 32-bit words)
_sig("alpgpu_packed_capacity_f32", _u64, _u64)
_sig("alpgpu_exc_capacity_f32", _u64, _u64)
_sig("alpgpu_decode_f32", _int, _vp, C.POINTER(CColumn), _vp)
_sig("alpgpu_decode_sum_f32", _int, _vp, C.POINTER(CColumn), _vp)
_sig("alpgpu_decode_count_range_f32", _int, _vp, C.POINTER(CColumn), C.c_float, C.c_float, _vp)
_sig("alpgpu_rowgroup_init_f32", _int, _vp, _vp, _u64, C.POINTER(CColumn))
_sig("alpgpu_encode_vectors_f32", _int, _vp, _vp, _u64, C.POINTER(CColumn))
_sig("alpgpu_encode_f32", _int, _vp, _vp, _u64, C.POINTER(CColumn))
_sig("alpgpu_state_from_samples_f32", _int, _vp, _vp, C.c_uint32, _vp)
_sig("alpgpu_rd_state_from_samples_f32", _int, _vp, _vp, C.c_uint32, _vp)
_sig("alpgpu_state_from_samples_f64", _int, _vp, _vp, C.c_uint32, _vp)
_sig("alpgpu_rd_state_from_samples_f64", _int, _vp, _vp, C.c_uint32, _vp)
_sig("alpgpu_rd_dictionary_for_cut_f64", _int, _vp, _vp, C.c_uint32, C.c_uint8, _vp, _vp)
_sig("alpgpu_rd_dictionary_for_cut_f32", _int, _vp, _vp, C.c_uint32, C.c_uint8, _vp, _vp)
_sig("alpgpu_pad_tail_f32", _int, _vp, _vp, _u64)
_sig("alpgpu_column_to_blob_f32", _int, _vp, C.POINTER(CColumn), _u64, _vp, _u64, C.POINTER(_u64))
_sig("alpgpu_column_from_blob_f32", _int, _vp, _vp, _u64, C.POINTER(CColumn), C.POINTER(_u64))
_sig("alpgpu_ffor_i32", _int, _vp, _vp, _vp, _sz, _vp, _vp, _u64)
_sig("alpgpu_unffor_i32", _int, _vp, _vp, _sz, _vp, _vp, _vp, _u64)
_sig("alpgpu_falp_f32", _int, _vp, _vp, _sz, _vp, _vp, _vp, _vp, _vp, _u64)
_sig("alpgpu_decode_values_f32", _int, _vp, _vp, _vp, _vp, _vp, _u64)
_sig("alpgpu_patch_f32", _int, _vp, _vp, _vp, _vp, _sz, _vp, _u64)
_sig("alpgpu_encode_simdized_f32", _int, _vp, _vp, _vp, _vp, _sz, _vp, _vp, _vp, _vp, _u64)
_sig("alpgpu_encode_values_f32", _int, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp, _vp, _vp, _vp, _u64)
_sig("alpgpu_analyze_ffor_i32", _int, _vp, _vp, _vp, _vp, _u64)
_sig("alpgpu_rd_encode_vectors_f32", _int, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp, _vp, _vp, _u64)
_sig("alpgpu_rd_decode_vectors_f32", _int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp, _u64)


class AlpGpuError(RuntimeError):
    pass


def _check(rc: int, what: str):
    if rc != 0:
        raise AlpGpuError(f"{what} failed ({rc}): {lib.alpgpu_last_error().decode()}")


class Context:
    """One per device / process.  With use_torch_stream (default) every call is enqueued on torch's CURRENT stream of the
    device at the time of the call (so `with torch.cuda.stream(s):` works as for torch's own ops); set_stream() pins a
    stream instead."""

    def __init__(self, device: int = 0, use_torch_stream: bool = True):
        h = _vp()
        _check(lib.alpgpu_ctx_create(device, C.byref(h)), "alpgpu_ctx_create")
        self._h = h
        self.device = device
        self._follow_torch = bool(use_torch_stream)
        self._last_stream = None

    @property
    def h(self):
        """the context handle; re-points the context at torch's current stream first when it follows torch"""
        if self._follow_torch and self._h:
            import torch
            cur = torch.cuda.current_stream(self.device).cuda_stream
            if cur != self._last_stream:
                _check(lib.alpgpu_set_stream(self._h, _vp(cur)), "alpgpu_set_stream")
                self._last_stream = cur
        return self._h

    def close(self):
        if getattr(self, "_h", None):
            lib.alpgpu_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, stream_handle: int):
        """pin the context to this hipStream_t handle (it stops following torch's current stream)"""
        self._follow_torch = False
        _check(lib.alpgpu_set_stream(self._h, _vp(stream_handle)), "alpgpu_set_stream")

    def set_option(self, option: int, value: int):
        _check(lib.alpgpu_set_option(self.h, option, value), "alpgpu_set_option")

    def synchronize(self):
        _check(lib.alpgpu_synchronize(self.h), "alpgpu_synchronize")

    def compress_host(self, x_host, blob_host=None):
        """host tensor (CPU, float64 / float32, any length; page-locked for speed) -> uint8 CPU tensor holding the serialized column
        (include/alpgpu.h: alpgpu_compress_host_*).  blob_host: a CPU uint8 tensor to write into (e.g. page-locked), else one is made."""
        t = self._sfx(x_host)
        n = (x_host.numel() + VECTOR_SIZE - 1) // VECTOR_SIZE
        if blob_host is None:
            cap = int(lib.alpgpu_blob_size(n, lib.alpgpu_packed_capacity(n), lib.alpgpu_exc_capacity(n)))
            blob_host = torch.empty(cap, dtype=torch.uint8)
        w = _u64()
        _check(getattr(lib, "alpgpu_compress_host_" + t)(self.h, _vp(x_host.data_ptr()), x_host.numel(), _vp(blob_host.data_ptr()), blob_host.numel(), C.byref(w)),
               "alpgpu_compress_host_" + t)
        return blob_host[: w.value]

    def decompress_host(self, blob_host, out_host):
        """serialized column (CPU uint8 tensor) -> out_host (CPU tensor of the column's type, at least as long as the column); returns the value count"""
        t = self._sfx(out_host)
        nv = _u64()
        _check(getattr(lib, "alpgpu_decompress_host_" + t)(self.h, _vp(blob_host.data_ptr()), blob_host.numel(), _vp(out_host.data_ptr()), out_host.numel(), C.byref(nv)),
               "alpgpu_decompress_host_" + t)
        return int(nv.value)

    @staticmethod
    def compress_host_multi(ctxs, x_host, blob_host=None):
        """alpgpu_compress_host_multi_*: the host column cut into whole-rowgroup shards over the contexts (one per GPU, normally), one blob"""
        import torch
        t = "f64" if x_host.dtype == torch.float64 else "f32"
        n = (x_host.numel() + 1023) // 1024
        if blob_host is None:
            cap = int(lib.alpgpu_blob_size(n, getattr(lib, "alpgpu_packed_capacity" + ("" if t == "f64" else "_f32"))(n), getattr(lib, "alpgpu_exc_capacity" + ("" if t == "f64" else "_f32"))(n)))
            blob_host = torch.empty(cap, dtype=torch.uint8)
        arr = (_vp * len(ctxs))(*[c.h for c in ctxs])
        w = _u64()
        _check(getattr(lib, "alpgpu_compress_host_multi_" + t)(arr, len(ctxs), _vp(x_host.data_ptr()), x_host.numel(), _vp(blob_host.data_ptr()), blob_host.numel(), C.byref(w)),
               "alpgpu_compress_host_multi_" + t)
        return blob_host[: w.value]

    @staticmethod
    def decompress_host_multi(ctxs, blob_host, out_host):
        import torch
        t = "f64" if out_host.dtype == torch.float64 else "f32"
        arr = (_vp * len(ctxs))(*[c.h for c in ctxs])
        nv = _u64()
        _check(getattr(lib, "alpgpu_decompress_host_multi_" + t)(arr, len(ctxs), _vp(blob_host.data_ptr()), blob_host.numel(), _vp(out_host.data_ptr()), out_host.numel(), C.byref(nv)),
               "alpgpu_decompress_host_multi_" + t)
        return nv.value

    def decode_probe(self, col: "DeviceColumn", out):
        """decode_sum without the unpack arithmetic (include/alpgpu.h: alpgpu_debug_decode_probe_f64)"""
        _check(lib.alpgpu_debug_decode_probe_f64(self.h, C.byref(col.c), _vp(out.data_ptr())), "alpgpu_debug_decode_probe_f64")

    def traffic_probe(self, x, out, n_vectors: int, write_bytes_per_vector: int):
        """the single-pass encode's loads and stores without its arithmetic (include/alpgpu.h: alpgpu_debug_traffic_probe)"""
        _check(lib.alpgpu_debug_traffic_probe(self.h, _vp(x.data_ptr()), _vp(out.data_ptr()), n_vectors, write_bytes_per_vector), "alpgpu_debug_traffic_probe")

    def traffic_probe_with_search(self, x, out, n_vectors: int, write_bytes_per_vector: int, scratch: "DeviceColumn"):
        """the same probe with the encode's rowgroup search (over x, a double column) running beside it as beside alpgpu_encode_f64"""
        _check(lib.alpgpu_debug_traffic_probe_with_search(self.h, _vp(x.data_ptr()), _vp(out.data_ptr()), n_vectors, write_bytes_per_vector, C.byref(scratch.c)),
               "alpgpu_debug_traffic_probe_with_search")

    def decode_vectors_per_wg(self, col: "DeviceColumn") -> int:
        """the launch shape decode() would use for this column now (vectors per decode workgroup)"""
        return int(lib.alpgpu_decode_vectors_per_wg(self.h, C.byref(col.c), 1 if col.dtype == "f32" else 0))

    def decode_runs(self, col: "DeviceColumn") -> int:
        """launches decode() of this column would make now: 1, or the runs of regions of different kinds (OPT_DECODE_SEGMENTS; after column_totals / from_blob)"""
        return int((lib.alpgpu_decode_runs_f32 if col.dtype == "f32" else lib.alpgpu_decode_runs)(self.h, C.byref(col.c)))

    def read_ahead_batches(self) -> int:
        """batches of 64 vectors this context's read-aheads have read since it was created (debug counter; waits for the streams)"""
        n = _u64()
        _check(lib.alpgpu_debug_read_ahead_batches(self.h, C.byref(n)), "alpgpu_debug_read_ahead_batches")
        return n.value

    def forget(self, col: "DeviceColumn"):
        """what the context remembers about this column (segments, learned sizes) is dropped, as an encode into it would"""
        _check(lib.alpgpu_debug_forget_column(self.h, C.byref(col.c)), "alpgpu_debug_forget_column")

    def unhinted_plan(self) -> dict:
        """the device-side plan of this context's last unhinted decode (include/alpgpu.h: alpgpu_debug_unhinted_plan)"""
        w = (_u64 * 6)()
        _check(lib.alpgpu_debug_unhinted_plan(self.h, C.byref(w)), "alpgpu_debug_unhinted_plan")
        return {"shape": int(w[0]), "lead_min": int(w[1] & 0xFFFFFFFF), "lead_max": int(w[1] >> 32), "ps_per_vector": int(w[2] & 0xFFFFFFFF), "max_bits": int(w[2] >> 32),
                "packed_bytes": int(w[3]), "exceptions": int(w[4]), "rd_vectors": int(w[5])}

    def decode_reads_ahead(self, col: "DeviceColumn") -> bool:
        """decode() of this column would start the read-ahead beside the decode kernel (OPT_DECODE_READ_AHEAD)"""
        return int(lib.alpgpu_decode_reads_ahead(self.h, C.byref(col.c), 1 if col.dtype == "f32" else 0)) == 1

    def device_info(self) -> dict:
        name = C.create_string_buffer(128)
        cus, hbm = _int(), _u64()
        _check(lib.alpgpu_device_info(self.h, name, 128, C.byref(cus), C.byref(hbm)), "alpgpu_device_info")
        return {"name": name.value.decode(), "cu_count": cus.value, "hbm_bytes": hbm.value}

    # ---- whole-column path ------------------------------------------------------------------------
    # Every whole-column call dispatches on the value type: float64 tensors / DeviceColumn(dtype="f64") -> *_f64 entry
    # points, float32 / "f32" -> *_f32.
    @staticmethod
    def _sfx(x):
        import torch
        assert x.dtype in (torch.float64, torch.float32)
        return "f64" if x.dtype == torch.float64 else "f32"

    def _check_input(self, x, col):
        assert x.is_contiguous() and x.is_cuda and self._sfx(x) == col.dtype, "input tensor and column must have the same value type"
        assert x.numel() == col.n_vectors * VECTOR_SIZE, "input must hold exactly n_vectors * 1024 values"

    def _call(self, stem, sfx, *args):
        name = f"alpgpu_{stem}_{sfx}"
        _check(getattr(lib, name)(self.h, *args), name)

    def rowgroup_init(self, x, col: "DeviceColumn"):
        self._check_input(x, col)
        self._call("rowgroup_init", col.dtype, _vp(x.data_ptr()), col.n_vectors, C.byref(col.c))

    def encode_vectors(self, x, col: "DeviceColumn"):
        self._check_input(x, col)
        self._call("encode_vectors", col.dtype, _vp(x.data_ptr()), col.n_vectors, C.byref(col.c))

    def encode(self, x, col: "DeviceColumn" = None):
        """rowgroup init + vector encode of a device tensor of n_vectors*1024 doubles (or floats)"""
        if col is None:
            col = DeviceColumn(x.numel() // VECTOR_SIZE, self.device, dtype=self._sfx(x))
        self._check_input(x, col)
        self._call("encode", col.dtype, _vp(x.data_ptr()), col.n_vectors, C.byref(col.c))
        return col

    def state_from_samples(self, samples, state, rd_only: bool = False):
        """samples: device tensor of 1..288 first-level samples; state: 32-byte uint8 device tensor"""
        self._call("rd_state_from_samples" if rd_only else "state_from_samples", self._sfx(samples), _vp(samples.data_ptr()), samples.numel(),
                   _vp(state.data_ptr()))

    def rd_dictionary_for_cut(self, samples, right_bit_width: int, state, estimate):
        """rd_encoder::build_left_parts_dictionary for one cut: state (32-byte uint8 device tensor) and estimate (1 float64, device) are written"""
        self._call("rd_dictionary_for_cut", self._sfx(samples), _vp(samples.data_ptr()), samples.numel(), right_bit_width, _vp(state.data_ptr()), _vp(estimate.data_ptr()))

    # ---- tail padding + serialized container ------------------------------------------------------
    def pad_tail(self, x, n_values: int):
        """x: device tensor with room for ceil(n_values/1024)*1024 doubles; fills the incomplete last vector"""
        assert x.numel() >= (n_values + 1023) // 1024 * 1024
        self._call("pad_tail", self._sfx(x), _vp(x.data_ptr()), n_values)

    def to_blob(self, col: "DeviceColumn", n_values: int) -> np.ndarray:
        pb, eb, ov = self.column_totals(col)
        size = int(lib.alpgpu_blob_size(col.n_vectors, pb, eb))
        blob = np.zeros(size, np.uint8)
        w = _u64()
        fn = lib.alpgpu_column_to_blob if col.dtype == "f64" else lib.alpgpu_column_to_blob_f32
        _check(fn(self.h, C.byref(col.c), n_values, blob.ctypes.data_as(_vp), size, C.byref(w)), "alpgpu_column_to_blob")
        assert w.value == size
        return blob

    def from_blob(self, blob: np.ndarray):
        """-> (DeviceColumn, n_values); raises AlpGpuError on a malformed blob"""
        if blob.size < 64:
            raise AlpGpuError("blob shorter than its 64-byte header")
        hdr = np.frombuffer(blob[:64].tobytes(), dtype=np.uint64)
        n_vectors, packed_bytes, exc_bytes = int(hdr[3]), int(hdr[5]), int(hdr[6])
        if n_vectors > (1 << 40) or packed_bytes > (1 << 50) or exc_bytes > (1 << 50):
            raise AlpGpuError("blob header is implausible")
        dtype = "f32" if int(hdr[7]) == 4 else "f64"
        col = DeviceColumn(n_vectors, self.device, packed_capacity=packed_bytes + 1024, exc_capacity=exc_bytes + 64, dtype=dtype)
        nv = _u64()
        fn = lib.alpgpu_column_from_blob if dtype == "f64" else lib.alpgpu_column_from_blob_f32
        _check(fn(self.h, blob.ctypes.data_as(_vp), blob.size, C.byref(col.c), C.byref(nv)), "alpgpu_column_from_blob")
        return col, nv.value

    def column_totals(self, col: "DeviceColumn"):
        pb, eb, ov = _u64(), _u64(), _int()
        _check(lib.alpgpu_column_totals(self.h, C.byref(col.c), C.byref(pb), C.byref(eb), C.byref(ov)), "alpgpu_column_totals")
        return pb.value, eb.value, ov.value

    # ---- batch primitives (torch tensors on this device; shapes [n, 1024] unless noted) ---------------
    @staticmethod
    def _p(t):
        return _vp(t.data_ptr()) if t is not None else _vp(0)

    def ffor_i64(self, vals, packed, bw, base):
        _check(lib.alpgpu_ffor_i64(self.h, self._p(vals), self._p(packed), packed.shape[1], self._p(bw), self._p(base), vals.shape[0]), "alpgpu_ffor_i64")

    def unffor_i64(self, packed, out, bw, base):
        _check(lib.alpgpu_unffor_i64(self.h, self._p(packed), packed.shape[1], self._p(out), self._p(bw), self._p(base), out.shape[0]), "alpgpu_unffor_i64")

    def ffor_u16(self, vals, packed, bw, base=None):
        _check(lib.alpgpu_ffor_u16(self.h, self._p(vals), self._p(packed), packed.shape[1], self._p(bw), self._p(base), vals.shape[0]), "alpgpu_ffor_u16")

    def unffor_u16(self, packed, out, bw, base=None):
        _check(lib.alpgpu_unffor_u16(self.h, self._p(packed), packed.shape[1], self._p(out), self._p(bw), self._p(base), out.shape[0]), "alpgpu_unffor_u16")

    def ffor_u8(self, vals, packed, bw, base=None):
        _check(lib.alpgpu_ffor_u8(self.h, self._p(vals), self._p(packed), packed.shape[1], self._p(bw), self._p(base), vals.shape[0]), "alpgpu_ffor_u8")

    def unffor_u8(self, packed, out, bw, base=None):
        _check(lib.alpgpu_unffor_u8(self.h, self._p(packed), packed.shape[1], self._p(out), self._p(bw), self._p(base), out.shape[0]), "alpgpu_unffor_u8")

    def falp(self, packed, out, bw, base, fac, exp):
        _check(lib.alpgpu_falp_f64(self.h, self._p(packed), packed.shape[1], self._p(out), self._p(bw), self._p(base), self._p(fac), self._p(exp),
                                   out.shape[0]), "alpgpu_falp_f64")

    def decode_values(self, enc, out, fac, exp):
        _check(lib.alpgpu_decode_values_f64(self.h, self._p(enc), self._p(out), self._p(fac), self._p(exp), out.shape[0]), "alpgpu_decode_values_f64")

    def patch(self, out, exc, pos, cnt):
        _check(lib.alpgpu_patch_f64(self.h, self._p(out), self._p(exc), self._p(pos), exc.shape[1], self._p(cnt), out.shape[0]), "alpgpu_patch_f64")

    def encode_simdized(self, x, exc, pos, cnt, enc, fac, exp):
        _check(lib.alpgpu_encode_simdized_f64(self.h, self._p(x), self._p(exc), self._p(pos), exc.shape[1], self._p(cnt), self._p(enc),
                                              self._p(fac), self._p(exp), x.shape[0]), "alpgpu_encode_simdized_f64")

    def encode_values(self, x, states, state_idx, exc, pos, cnt, enc, fac, exp):
        _check(lib.alpgpu_encode_values_f64(self.h, self._p(x), self._p(states), self._p(state_idx), self._p(exc), self._p(pos), exc.shape[1],
                                            self._p(cnt), self._p(enc), self._p(fac), self._p(exp), x.shape[0]), "alpgpu_encode_values_f64")

    def encode_value(self, x, fac: int, exp: int, safe: bool = True):
        """alpgpu_encode_value_f64 / _f32 on a 1-d device tensor: the encoded integers (int64 / int32)"""
        import torch
        f64 = x.dtype == torch.float64
        out = torch.empty(x.numel(), dtype=torch.int64 if f64 else torch.int32, device=x.device)
        fn = lib.alpgpu_encode_value_f64 if f64 else lib.alpgpu_encode_value_f32
        _check(fn(self.h, self._p(x), self._p(out), fac, exp, 1 if safe else 0, x.numel()), "alpgpu_encode_value")
        return out

    def analyze_ffor(self, enc, bw, base):
        _check(lib.alpgpu_analyze_ffor_i64(self.h, self._p(enc), self._p(bw), self._p(base), enc.shape[0]), "alpgpu_analyze_ffor_i64")

    def rd_encode_vectors(self, x, states, state_idx, exc, pos, cnt, right, left):
        _check(lib.alpgpu_rd_encode_vectors_f64(self.h, self._p(x), self._p(states), self._p(state_idx), self._p(exc), self._p(pos), exc.shape[1],
                                                self._p(cnt), self._p(right), self._p(left), x.shape[0]), "alpgpu_rd_encode_vectors_f64")

    def rd_decode_vectors(self, out, right, left, states, state_idx, exc, pos, cnt):
        _check(lib.alpgpu_rd_decode_vectors_f64(self.h, self._p(out), self._p(right), self._p(left), self._p(states), self._p(state_idx),
                                                self._p(exc), self._p(pos), exc.shape[1], self._p(cnt), out.shape[0]), "alpgpu_rd_decode_vectors_f64")

    def decode_count_range(self, col: "DeviceColumn", lo: float, hi: float, out=None):
        """per-vector count of decoded values in [lo, hi] without materialising them (alpgpu_decode_count_range_f64 / _f32)"""
        import torch
        if out is None:
            out = torch.empty(col.n_vectors, dtype=torch.int32, device=f"cuda:{self.device}")
        fn = lib.alpgpu_decode_count_range_f64 if col.dtype == "f64" else lib.alpgpu_decode_count_range_f32
        _check(fn(self.h, C.byref(col.c), lo, hi, _vp(out.data_ptr())), "alpgpu_decode_count_range")
        return out

    def column_sum(self, col: "DeviceColumn", out=None):
        """the whole column's total in the documented tree order (alpgpu_column_sum_f64 / _f32): a 1-element float64 tensor"""
        import torch
        if out is None:
            out = torch.empty(1, dtype=torch.float64, device=col.vectors.device)
        fn = lib.alpgpu_column_sum_f64 if col.dtype == "f64" else lib.alpgpu_column_sum_f32
        _check(fn(self.h, C.byref(col.c), _vp(out.data_ptr())), "alpgpu_column_sum")
        return out

    def tree_sum(self, x, out=None):
        """alpgpu_tree_sum_f64 over a float64 device tensor"""
        import torch
        if out is None:
            out = torch.empty(1, dtype=torch.float64, device=x.device)
        _check(lib.alpgpu_tree_sum_f64(self.h, _vp(x.data_ptr()), x.numel(), _vp(out.data_ptr())), "alpgpu_tree_sum_f64")
        return out

    def column_validate(self, col: "DeviceColumn"):
        """alpgpu_column_validate: None when every descriptor is well-formed, else the index of the first bad vector"""
        bad = _u64(0)
        rc = lib.alpgpu_column_validate(self.h, C.byref(col.c), 8 if col.dtype == "f64" else 4, C.byref(bad))
        if rc == 0:
            return None
        if rc != -2:
            _check(rc, "alpgpu_column_validate")
        return int(bad.value)

    def decode_sum(self, col: "DeviceColumn", out=None):
        """per-vector sums (float64) of the decoded values without materialising them (alpgpu_decode_sum_f64 / _f32)"""
        import torch
        if out is None:
            out = torch.empty(col.n_vectors, dtype=torch.float64, device=f"cuda:{self.device}")
        fn = lib.alpgpu_decode_sum_f64 if col.dtype == "f64" else lib.alpgpu_decode_sum_f32
        _check(fn(self.h, C.byref(col.c), _vp(out.data_ptr())), "alpgpu_decode_sum")
        return out

    def decode(self, col: "DeviceColumn", out=None):
        import torch
        tdt = torch.float64 if col.dtype == "f64" else torch.float32
        if out is None:
            out = torch.empty(col.n_vectors * VECTOR_SIZE, dtype=tdt, device=f"cuda:{self.device}")
        assert out.is_contiguous() and out.numel() >= col.n_vectors * VECTOR_SIZE and out.dtype == tdt
        self._call("decode", col.dtype, C.byref(col.c), _vp(out.data_ptr()))
        return out

    # ---- batch primitives, 32-bit words (float) --------------------------------------------------------
    def ffor_i32(self, vals, packed, bw, base):
        _check(lib.alpgpu_ffor_i32(self.h, self._p(vals), self._p(packed), packed.shape[1], self._p(bw), self._p(base), vals.shape[0]), "alpgpu_ffor_i32")

    def unffor_i32(self, packed, out, bw, base):
        _check(lib.alpgpu_unffor_i32(self.h, self._p(packed), packed.shape[1], self._p(out), self._p(bw), self._p(base), out.shape[0]), "alpgpu_unffor_i32")

    def falp_f32(self, packed, out, bw, base, fac, exp):
        _check(lib.alpgpu_falp_f32(self.h, self._p(packed), packed.shape[1], self._p(out), self._p(bw), self._p(base), self._p(fac), self._p(exp),
                                   out.shape[0]), "alpgpu_falp_f32")

    def decode_values_f32(self, enc, out, fac, exp):
        _check(lib.alpgpu_decode_values_f32(self.h, self._p(enc), self._p(out), self._p(fac), self._p(exp), out.shape[0]), "alpgpu_decode_values_f32")

    def patch_f32(self, out, exc, pos, cnt):
        _check(lib.alpgpu_patch_f32(self.h, self._p(out), self._p(exc), self._p(pos), exc.shape[1], self._p(cnt), out.shape[0]), "alpgpu_patch_f32")

    def encode_simdized_f32(self, x, exc, pos, cnt, enc, fac, exp):
        _check(lib.alpgpu_encode_simdized_f32(self.h, self._p(x), self._p(exc), self._p(pos), exc.shape[1], self._p(cnt), self._p(enc),
                                              self._p(fac), self._p(exp), x.shape[0]), "alpgpu_encode_simdized_f32")

    def encode_values_f32(self, x, states, state_idx, exc, pos, cnt, enc, fac, exp):
        _check(lib.alpgpu_encode_values_f32(self.h, self._p(x), self._p(states), self._p(state_idx), self._p(exc), self._p(pos), exc.shape[1],
                                            self._p(cnt), self._p(enc), self._p(fac), self._p(exp), x.shape[0]), "alpgpu_encode_values_f32")

    def analyze_ffor_i32(self, enc, bw, base):
        _check(lib.alpgpu_analyze_ffor_i32(self.h, self._p(enc), self._p(bw), self._p(base), enc.shape[0]), "alpgpu_analyze_ffor_i32")

    def rd_encode_vectors_f32(self, x, states, state_idx, exc, pos, cnt, right, left):
        _check(lib.alpgpu_rd_encode_vectors_f32(self.h, self._p(x), self._p(states), self._p(state_idx), self._p(exc), self._p(pos), exc.shape[1],
                                                self._p(cnt), self._p(right), self._p(left), x.shape[0]), "alpgpu_rd_encode_vectors_f32")

    def rd_decode_vectors_f32(self, out, right, left, states, state_idx, exc, pos, cnt):
        _check(lib.alpgpu_rd_decode_vectors_f32(self.h, self._p(out), self._p(right), self._p(left), self._p(states), self._p(state_idx),
                                                self._p(exc), self._p(pos), exc.shape[1], self._p(cnt), out.shape[0]), "alpgpu_rd_decode_vectors_f32")


class DeviceColumn:
    """A compressed column in HBM (struct alpgpu_column) whose buffers are torch uint8 tensors."""

    def __init__(self, n_vectors: int, device: int = 0, packed_capacity: int | None = None,
                 exc_capacity: int | None = None, dtype: str = "f64", rd_order: bool = True):
        import torch
        assert dtype in ("f64", "f32")
        dev = f"cuda:{device}"
        self.dtype = dtype
        self.n_vectors = int(n_vectors)
        self.n_rowgroups = (self.n_vectors + ROWGROUP_VECTORS - 1) // ROWGROUP_VECTORS
        pcap = lib.alpgpu_packed_capacity if dtype == "f64" else lib.alpgpu_packed_capacity_f32
        ecap = lib.alpgpu_exc_capacity if dtype == "f64" else lib.alpgpu_exc_capacity_f32
        pc = int(pcap(self.n_vectors)) if packed_capacity is None else int(packed_capacity)
        ec = int(ecap(self.n_vectors)) if exc_capacity is None else int(exc_capacity)
        self.rowgroups = torch.zeros(max(1, self.n_rowgroups) * 32, dtype=torch.uint8, device=dev)
        self.vectors = torch.zeros(max(1, self.n_vectors) * 32, dtype=torch.uint8, device=dev)
        self.packed = torch.zeros(pc, dtype=torch.uint8, device=dev)
        self.exc = torch.zeros(ec, dtype=torch.uint8, device=dev)
        self.totals = torch.zeros(8, dtype=torch.int64, device=dev)
        # optional ALP_RD sorted-order table (exception-slot indices identical to the reference's); rd_order=False leaves it out
        self.rd_order = torch.zeros(max(1, self.n_rowgroups) * RD_ORDER_STRIDE, dtype=torch.int16, device=dev) if rd_order else None
        self.c = CColumn(self.n_vectors, self.n_rowgroups, self.rowgroups.data_ptr(), self.vectors.data_ptr(),
                         self.packed.data_ptr(), pc, self.exc.data_ptr(), ec, self.totals.data_ptr(), 0, 0,
                         self.rd_order.data_ptr() if rd_order else None)

    @classmethod
    def from_host(cls, rowgroups: np.ndarray, vectors: np.ndarray, packed: np.ndarray, exc: np.ndarray, device: int = 0, dtype: str = "f64"):
        """Upload host-side records/streams (numpy; see ROWGROUP_DTYPE / VECTOR_DTYPE)."""
        import torch
        n = vectors.size
        col = cls(n, device, packed_capacity=packed.size + 1024, exc_capacity=exc.size + 64, dtype=dtype)
        col.rowgroups[: rowgroups.size * 32] = torch.from_numpy(rowgroups.view(np.uint8).reshape(-1)).to(col.rowgroups.device)
        col.vectors[: n * 32] = torch.from_numpy(vectors.view(np.uint8).reshape(-1)).to(col.vectors.device)
        col.packed[: packed.size] = torch.from_numpy(packed).to(col.packed.device)
        col.exc[: exc.size] = torch.from_numpy(exc).to(col.exc.device)
        col.totals[0] = packed.size
        col.totals[1] = exc.size
        col.c.packed_bytes_hint, col.c.exc_bytes_hint = packed.size, exc.size
        col.c.alp_rd_rowgroups_hint = 1 + int((rowgroups["scheme"] == SCHEME_ALP_RD).sum())
        return col

    def to_host(self):
        rg = self.rowgroups.cpu().numpy().view(ROWGROUP_DTYPE)[: self.n_rowgroups]
        vec = self.vectors.cpu().numpy().view(VECTOR_DTYPE)[: self.n_vectors]
        tot = self.totals.cpu().numpy()
        packed = self.packed[: int(tot[0])].cpu().numpy()
        exc = self.exc[: int(tot[1])].cpu().numpy()
        return rg, vec, packed, exc
