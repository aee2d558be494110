"""ctypes front-end for the two CPU checkers under oracle/ — TEST INFRASTRUCTURE ONLY.

* ``oracle/libalp_oracle.so``      — our plain-C restatement (oracle/alp_oracle.c), prefix ``alpo_``
* ``oracle/_ref/libalp_ref.so``    — the real reference compiled in place from /root/reference with the
                                     extern "C" shim oracle/ref_harness.cpp, prefix ``ref_``

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product (alp_amd, libalpgpu.so) never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "libalp_oracle.so")
REF_SO = os.path.join(HERE, "_ref", "libalp_ref.so")
REF_AVX512_SO = os.path.join(HERE, "_ref", "libalp_ref_avx512.so")

VECTOR_SIZE = 1024
ROWGROUP_VECTORS = 100
SCHEME_ALP_RD, SCHEME_ALP = 1, 2


def build(ref: bool = True) -> None:
    """(Re)build the checkers.  The reference build is attempted only when /root/reference exists."""
    subprocess.check_call(["make", "-s", "-C", HERE, "oracle"])
    if ref:
        subprocess.check_call(["make", "-s", "-C", HERE, "ref"])


def _p(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


class _Base:
    prefix = ""

    def __init__(self, path: str):
        self.path = path
        self.lib = C.CDLL(path)

    def fn(self, name, restype=None):
        f = getattr(self.lib, self.prefix + name)
        f.restype = restype
        return f

    # ---- whole column ---------------------------------------------------------------------------
    def encode_column(self, col: np.ndarray) -> dict:
        col = np.ascontiguousarray(col, dtype=np.float64)
        assert col.size % VECTOR_SIZE == 0
        n = col.size // VECTOR_SIZE
        nrg = (n + ROWGROUP_VECTORS - 1) // ROWGROUP_VECTORS
        o = dict(
            scheme=np.zeros(n, np.uint8), e=np.zeros(n, np.uint8), f=np.zeros(n, np.uint8), bw=np.zeros(n, np.uint8),
            lbw=np.zeros(n, np.uint8), base=np.zeros(n, np.int64), exc_cnt=np.zeros(n, np.uint16),
            packed=np.zeros((n, 1024), np.int64), packed_left=np.zeros((n, 1024), np.uint16),
            exc=np.zeros((n, 1024), np.float64), pos=np.zeros((n, 1024), np.uint16),
            dict=np.zeros((nrg, 8), np.uint16), dict_size=np.zeros(nrg, np.uint8), k=np.zeros(nrg, np.uint8),
            combos=np.zeros((nrg, 10), np.int32),
        )
        self.fn("encode_column")(
            _p(col), C.c_size_t(n), _p(o["scheme"]), _p(o["e"]), _p(o["f"]), _p(o["bw"]), _p(o["lbw"]), _p(o["base"]),
            _p(o["exc_cnt"]), _p(o["packed"]), _p(o["packed_left"]), _p(o["exc"]), _p(o["pos"]), _p(o["dict"]),
            _p(o["dict_size"]), _p(o["k"]), _p(o["combos"]))
        return o


class Oracle(_Base):
    """Plain-C restatement (kind "port")."""
    prefix = "alpo_"

    def __init__(self, path: str = ORACLE_SO):
        if not os.path.exists(path):
            build(ref=False)
        super().__init__(path)

    def decode_column(self, o: dict) -> np.ndarray:
        n = o["scheme"].size
        out = np.empty(n * VECTOR_SIZE, np.float64)
        self.fn("decode_column")(
            C.c_size_t(n), _p(o["scheme"]), _p(o["e"]), _p(o["f"]), _p(o["bw"]), _p(o["lbw"]), _p(o["base"]),
            _p(o["exc_cnt"]), _p(o["packed"]), _p(o["packed_left"]), _p(o["exc"]), _p(o["pos"]), _p(o["dict"]), _p(out))
        return out

    # ---- primitives -----------------------------------------------------------------------------
    def ffor_u64(self, vals: np.ndarray, bw: int, base: int) -> np.ndarray:
        vals = np.ascontiguousarray(vals).view(np.uint64)
        out = np.zeros(1024, np.uint64)
        self.fn("ffor_u64")(_p(vals), _p(out), C.c_int(bw), C.c_uint64(base & (2**64 - 1)))
        return out

    def unffor_u64(self, packed: np.ndarray, bw: int, base: int) -> np.ndarray:
        packed = np.ascontiguousarray(packed).view(np.uint64)
        out = np.zeros(1024, np.uint64)
        self.fn("unffor_u64")(_p(packed), _p(out), C.c_int(bw), C.c_uint64(base & (2**64 - 1)))
        return out

    def ffor_u16(self, vals: np.ndarray, bw: int, base: int = 0) -> np.ndarray:
        vals = np.ascontiguousarray(vals, np.uint16)
        out = np.zeros(1024, np.uint16)
        self.fn("ffor_u16")(_p(vals), _p(out), C.c_int(bw), C.c_uint16(base))
        return out

    def unffor_u16(self, packed: np.ndarray, bw: int, base: int = 0) -> np.ndarray:
        packed = np.ascontiguousarray(packed, np.uint16)
        out = np.zeros(1024, np.uint16)
        self.fn("unffor_u16")(_p(packed), _p(out), C.c_int(bw), C.c_uint16(base))
        return out

    def ffor_u8(self, vals, bw, base=0):
        vals = np.ascontiguousarray(vals, np.uint8)
        out = np.zeros(1024, np.uint8)
        self.fn("ffor_u8")(_p(vals), _p(out), C.c_int(bw), C.c_uint8(base))
        return out

    def unffor_u8(self, packed, bw, base=0):
        packed = np.ascontiguousarray(packed, np.uint8)
        out = np.zeros(1024, np.uint8)
        self.fn("unffor_u8")(_p(packed), _p(out), C.c_int(bw), C.c_uint8(base))
        return out

    def falp(self, packed: np.ndarray, bw: int, base: int, fac: int, exp: int) -> np.ndarray:
        packed = np.ascontiguousarray(packed).view(np.uint64)
        out = np.zeros(1024, np.float64)
        self.fn("falp")(_p(packed), _p(out), C.c_int(bw), C.c_uint64(base & (2**64 - 1)), C.c_int(fac), C.c_int(exp))
        return out

    def encode_simdized(self, vec: np.ndarray, fac: int, exp: int):
        vec = np.ascontiguousarray(vec, np.float64)
        exc = np.zeros(1024, np.float64)
        pos = np.zeros(1024, np.uint16)
        cnt = np.zeros(1, np.uint16)
        enc = np.zeros(1024, np.int64)
        self.fn("encode_simdized")(_p(vec), _p(exc), _p(pos), _p(cnt), _p(enc), C.c_int(fac), C.c_int(exp))
        return enc, exc, pos, int(cnt[0])

    def analyze_ffor(self, enc: np.ndarray):
        enc = np.ascontiguousarray(enc, np.int64)
        bw = np.zeros(1, np.uint8)
        base = np.zeros(1, np.int64)
        self.fn("analyze_ffor")(_p(enc), _p(bw), _p(base))
        return int(bw[0]), int(base[0])

    def time_falp_column(self, packed, stride_words, bw, e, f, base, exc_cnt, exc, pos, exc_stride, n, out, reps):
        return self.fn("time_falp_column", C.c_double)(
            _p(packed), C.c_size_t(stride_words), _p(bw), _p(e), _p(f), _p(base), _p(exc_cnt), _p(exc), _p(pos),
            C.c_size_t(exc_stride), C.c_size_t(n), _p(out), C.c_int(reps))


class Reference(_Base):
    """The real cwida/ALP code (kind "reference").  Present only if oracle/_ref was built (in the build
    container, from /root/reference) and shipped with the snapshot."""
    prefix = "ref_"

    def __init__(self, path: str = REF_SO):
        super().__init__(path)

    @staticmethod
    def available(path: str = REF_SO) -> bool:
        return os.path.exists(path)

    def ffor_u64(self, vals, bw, base):
        vals = np.ascontiguousarray(vals).view(np.uint64)
        out = np.zeros(1024, np.uint64)
        b = np.array([base & (2**64 - 1)], np.uint64)
        self.fn("ffor_u64")(_p(vals), _p(out), C.c_uint8(bw), _p(b))
        return out

    def unffor_u64(self, packed, bw, base):
        packed = np.ascontiguousarray(packed).view(np.uint64)
        out = np.zeros(1024, np.uint64)
        b = np.array([base & (2**64 - 1)], np.uint64)
        self.fn("unffor_u64")(_p(packed), _p(out), C.c_uint8(bw), _p(b))
        return out

    def ffor_u16(self, vals, bw, base=0):
        vals = np.ascontiguousarray(vals, np.uint16)
        out = np.zeros(1024, np.uint16)
        b = np.array([base], np.uint16)
        self.fn("ffor_u16")(_p(vals), _p(out), C.c_uint8(bw), _p(b))
        return out

    def unffor_u16(self, packed, bw, base=0):
        packed = np.ascontiguousarray(packed, np.uint16)
        out = np.zeros(1024, np.uint16)
        b = np.array([base], np.uint16)
        self.fn("unffor_u16")(_p(packed), _p(out), C.c_uint8(bw), _p(b))
        return out

    def ffor_u8(self, vals, bw, base=0):
        vals = np.ascontiguousarray(vals, np.uint8)
        out = np.zeros(1024, np.uint8)
        b = np.array([base], np.uint8)
        self.fn("ffor_u8")(_p(vals), _p(out), C.c_uint8(bw), _p(b))
        return out

    def unffor_u8(self, packed, bw, base=0):
        packed = np.ascontiguousarray(packed, np.uint8)
        out = np.zeros(1024, np.uint8)
        b = np.array([base], np.uint8)
        self.fn("unffor_u8")(_p(packed), _p(out), C.c_uint8(bw), _p(b))
        return out

    def falp(self, packed, bw, base, fac, exp):
        packed = np.ascontiguousarray(packed).view(np.int64)
        out = np.zeros(1024, np.float64)
        b = np.array([base & (2**64 - 1)], np.uint64).view(np.int64)
        self.fn("falp")(_p(packed), _p(out), C.c_uint8(bw), _p(b), C.c_uint8(fac), C.c_uint8(exp))
        return out

    def unffor_decode(self, packed, bw, base, fac, exp):
        """The unfused path benchmarks/benchmark.cpp:129-131 uses: unffor then decoder::decode."""
        packed = np.ascontiguousarray(packed).view(np.int64)
        tmp = np.zeros(1024, np.int64)
        out = np.zeros(1024, np.float64)
        b = np.array([base & (2**64 - 1)], np.uint64).view(np.int64)
        self.fn("unffor_i64")(_p(packed), _p(tmp), C.c_uint8(bw), _p(b))
        self.fn("decode")(_p(tmp), C.c_uint8(fac), C.c_uint8(exp), _p(out))
        return out

    def encode_simdized(self, vec, fac, exp):
        vec = np.ascontiguousarray(vec, np.float64)
        exc = np.zeros(1024, np.float64)
        pos = np.zeros(1024, np.uint16)
        cnt = np.zeros(1, np.uint16)
        enc = np.zeros(1024, np.int64)
        self.fn("encode_simdized")(_p(vec), _p(exc), _p(pos), _p(cnt), _p(enc), C.c_uint8(fac), C.c_uint8(exp))
        return enc, exc, pos, int(cnt[0])

    def analyze_ffor(self, enc):
        enc = np.ascontiguousarray(enc, np.int64)
        bw = np.zeros(1, np.uint8)
        base = np.zeros(1, np.int64)
        self.fn("analyze_ffor")(_p(enc), _p(bw), _p(base))
        return int(bw[0]), int(base[0])

    def time_falp_column(self, packed, stride_words, bw, e, f, base, exc_cnt, exc, pos, exc_stride, n, out, reps):
        return self.fn("time_falp_column", C.c_double)(
            _p(packed), C.c_size_t(stride_words), _p(bw), _p(e), _p(f), _p(base), _p(exc_cnt), _p(exc), _p(pos),
            C.c_size_t(exc_stride), C.c_size_t(n), _p(out), C.c_int(reps))

    def time_encode_column(self, col, n, reps=1):
        scratch = np.zeros(1024, np.int64)
        s = np.zeros(1, np.uint64)
        t = self.fn("time_encode_column", C.c_double)(_p(col), C.c_size_t(n), _p(scratch), C.c_int(reps), _p(s))
        return t, int(s[0])


# ===================================================================================================
# single precision: alp_oracle_f32.c (prefix alpof_) and the reference's float instantiation (reff_)
# ===================================================================================================
def _encode_column_f32(fn, col: np.ndarray) -> dict:
    col = np.ascontiguousarray(col, dtype=np.float32)
    assert col.size % VECTOR_SIZE == 0
    n = col.size // VECTOR_SIZE
    nrg = (n + ROWGROUP_VECTORS - 1) // ROWGROUP_VECTORS
    o = dict(
        scheme=np.zeros(n, np.uint8), e=np.zeros(n, np.uint8), f=np.zeros(n, np.uint8), bw=np.zeros(n, np.uint8),
        lbw=np.zeros(n, np.uint8), base=np.zeros(n, np.int64), exc_cnt=np.zeros(n, np.uint16),
        packed=np.zeros((n, 1024), np.int32), packed_left=np.zeros((n, 1024), np.uint16),
        exc=np.zeros((n, 1024), np.float32), pos=np.zeros((n, 1024), np.uint16),
        dict=np.zeros((nrg, 8), np.uint16), dict_size=np.zeros(nrg, np.uint8), k=np.zeros(nrg, np.uint8),
        combos=np.zeros((nrg, 10), np.int32),
    )
    fn(_p(col), C.c_size_t(n), _p(o["scheme"]), _p(o["e"]), _p(o["f"]), _p(o["bw"]), _p(o["lbw"]), _p(o["base"]),
       _p(o["exc_cnt"]), _p(o["packed"]), _p(o["packed_left"]), _p(o["exc"]), _p(o["pos"]), _p(o["dict"]),
       _p(o["dict_size"]), _p(o["k"]), _p(o["combos"]))
    return o


class OracleF32(_Base):
    """Plain-C restatement of the float path (oracle/alp_oracle_f32.c)."""
    prefix = "alpof_"

    def __init__(self, path: str = ORACLE_SO):
        if not os.path.exists(path):
            build(ref=False)
        super().__init__(path)

    def encode_column(self, col):
        return _encode_column_f32(self.fn("encode_column"), col)

    def decode_column(self, o: dict) -> np.ndarray:
        n = o["scheme"].size
        out = np.empty(n * VECTOR_SIZE, np.float32)
        self.fn("decode_column")(
            C.c_size_t(n), _p(o["scheme"]), _p(o["e"]), _p(o["f"]), _p(o["bw"]), _p(o["lbw"]), _p(o["base"]),
            _p(o["exc_cnt"]), _p(o["packed"]), _p(o["packed_left"]), _p(o["exc"]), _p(o["pos"]), _p(o["dict"]), _p(out))
        return out

    def ffor_u32(self, vals, bw, base=0):
        vals = np.ascontiguousarray(vals).view(np.uint32)
        out = np.zeros(1024, np.uint32)
        self.fn("ffor_u32")(_p(vals), _p(out), C.c_int(bw), C.c_uint32(base & 0xFFFFFFFF))
        return out

    def unffor_u32(self, packed, bw, base=0):
        packed = np.ascontiguousarray(packed).view(np.uint32)
        out = np.zeros(1024, np.uint32)
        self.fn("unffor_u32")(_p(packed), _p(out), C.c_int(bw), C.c_uint32(base & 0xFFFFFFFF))
        return out

    def falp(self, packed, bw, base, fac, exp):
        packed = np.ascontiguousarray(packed).view(np.uint32)
        out = np.zeros(1024, np.float32)
        self.fn("falp")(_p(packed), _p(out), C.c_int(bw), C.c_uint32(base & 0xFFFFFFFF), C.c_int(fac), C.c_int(exp))
        return out

    def encode_simdized(self, vec, fac, exp):
        vec = np.ascontiguousarray(vec, np.float32)
        exc, pos, cnt, enc = np.zeros(1024, np.float32), np.zeros(1024, np.uint16), np.zeros(1, np.uint16), np.zeros(1024, np.int32)
        self.fn("encode_simdized")(_p(vec), _p(exc), _p(pos), _p(cnt), _p(enc), C.c_int(fac), C.c_int(exp))
        return enc, exc, pos, int(cnt[0])

    def analyze_ffor(self, enc):
        enc = np.ascontiguousarray(enc, np.int32)
        bw, base = np.zeros(1, np.uint8), np.zeros(1, np.int32)
        self.fn("analyze_ffor")(_p(enc), _p(bw), _p(base))
        return int(bw[0]), int(base[0])

    def encode_value(self, v, fac, exp):
        return self.fn("encode_value", C.c_int32)(C.c_float(v), C.c_int(fac), C.c_int(exp))

    def decode_value(self, enc, fac, exp):
        return self.fn("decode_value", C.c_float)(C.c_int32(enc), C.c_int(fac), C.c_int(exp))


class ReferenceF32(_Base):
    """The real reference's float instantiation (ref_harness.cpp, reff_*)."""
    prefix = "reff_"

    def __init__(self, path: str = REF_SO):
        super().__init__(path)

    @staticmethod
    def available(path: str = REF_SO) -> bool:
        if not os.path.exists(path):
            return False
        try:
            getattr(C.CDLL(path), "reff_encode_column")
            return True
        except AttributeError:
            return False

    def encode_column(self, col):
        return _encode_column_f32(self.fn("encode_column"), col)

    def ffor_u32(self, vals, bw, base=0):
        vals = np.ascontiguousarray(vals).view(np.uint32)
        out = np.zeros(1024, np.uint32)
        b = np.array([base & 0xFFFFFFFF], np.uint32)
        self.fn("ffor_u32")(_p(vals), _p(out), C.c_uint8(bw), _p(b))
        return out

    def unffor_u32(self, packed, bw, base=0):
        packed = np.ascontiguousarray(packed).view(np.uint32)
        out = np.zeros(1024, np.uint32)
        b = np.array([base & 0xFFFFFFFF], np.uint32)
        self.fn("unffor_u32")(_p(packed), _p(out), C.c_uint8(bw), _p(b))
        return out

    def falp(self, packed, bw, base, fac, exp):
        packed = np.ascontiguousarray(packed).view(np.int32)
        out = np.zeros(1024, np.float32)
        b = np.array([base & 0xFFFFFFFF], np.uint32).view(np.int32)
        self.fn("falp")(_p(packed), _p(out), C.c_uint8(bw), _p(b), C.c_uint8(fac), C.c_uint8(exp))
        return out

    def unffor_decode(self, packed, bw, base, fac, exp):
        packed = np.ascontiguousarray(packed).view(np.int32)
        out = np.zeros(1024, np.float32)
        b = np.array([base & 0xFFFFFFFF], np.uint32).view(np.int32)
        self.fn("unffor_decode")(_p(packed), _p(out), C.c_uint8(bw), _p(b), C.c_uint8(fac), C.c_uint8(exp))
        return out

    def encode_simdized(self, vec, fac, exp):
        vec = np.ascontiguousarray(vec, np.float32)
        exc, pos, cnt, enc = np.zeros(1024, np.float32), np.zeros(1024, np.uint16), np.zeros(1, np.uint16), np.zeros(1024, np.int32)
        self.fn("encode_simdized")(_p(vec), _p(exc), _p(pos), _p(cnt), _p(enc), C.c_uint8(fac), C.c_uint8(exp))
        return enc, exc, pos, int(cnt[0])

    def analyze_ffor(self, enc):
        enc = np.ascontiguousarray(enc, np.int32)
        bw, base = np.zeros(1, np.uint8), np.zeros(1, np.int32)
        self.fn("analyze_ffor")(_p(enc), _p(bw), _p(base))
        return int(bw[0]), int(base[0])

    def encode_value(self, v, fac, exp):
        return self.fn("encode_value", C.c_int32)(C.c_float(v), C.c_uint8(fac), C.c_uint8(exp))

    def decode_value(self, enc, fac, exp):
        return self.fn("decode_value", C.c_float)(C.c_int32(enc), C.c_uint8(fac), C.c_uint8(exp))
