"""CPU: libalpgpu.so loads without a GPU and exports every symbol include/alpgpu.h declares; with no device
the only thing it does is fail loudly (no CPU fallback)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "alpgpu.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(alpgpu_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from alp_amd import capi
    names = declared_symbols()
    assert len(names) >= 10
    missing = [n for n in names if not hasattr(capi.lib, n)]
    assert not missing, f"declared in include/alpgpu.h but not exported by libalpgpu.so: {missing}"


def test_record_sizes_match_header():
    from alp_amd import capi
    assert capi.ROWGROUP_DTYPE.itemsize == 32 and capi.VECTOR_DTYPE.itemsize == 32
    assert ctypes.sizeof(capi.CColumn) == 104  # ABI version 3: + alp_rd_rowgroups_hint (version 2: + d_rd_order, 96)
    assert capi.lib.alpgpu_abi_version() == 3


def test_no_cpu_fallback_without_device():
    import torch
    from alp_amd import capi
    if torch.cuda.is_available():
        return
    h = ctypes.c_void_p()
    rc = capi.lib.alpgpu_ctx_create(0, ctypes.byref(h))
    assert rc == -1 and not h.value, "without a GPU the library must refuse to create a context"
    assert b"no CPU fallback" in capi.lib.alpgpu_last_error() or b"no HIP device" in capi.lib.alpgpu_last_error()
