"""Host-side (numpy) conversion between the checkers' fixed-stride encode output (oracle/pyoracle.py
``encode_column``) and the compact HBM column layout of include/alpgpu.h.  Test infrastructure."""
import numpy as np

from alp_amd.capi import ROWGROUP_DTYPE, VECTOR_DTYPE, SCHEME_ALP, SCHEME_ALP_RD


def record_sizes(scheme, bw, lbw, cnt, value_bytes=8):
    scheme = scheme.astype(np.int64); bw = bw.astype(np.int64); lbw = lbw.astype(np.int64); cnt = cnt.astype(np.int64)
    packed = np.where(scheme == SCHEME_ALP, 128 * bw, 128 * (bw + lbw))
    excb = np.where(scheme == SCHEME_ALP, (value_bytes + 2) * cnt, 4 * cnt)
    excb = (excb + 7) // 8 * 8
    return packed, excb


def compact(o, value_bytes=8):
    """oracle output dict -> (rowgroups[ROWGROUP_DTYPE], vectors[VECTOR_DTYPE], packed u8, exc u8); value_bytes = 4 for the float
    checkers' output (32-bit packed words / exception values)"""
    W = value_bytes
    n = o["scheme"].size
    nrg = (n + 99) // 100
    rg = np.zeros(nrg, ROWGROUP_DTYPE)
    for r in range(nrg):
        v0 = r * 100
        rg["scheme"][r] = o["scheme"][v0]
        if o["scheme"][v0] == SCHEME_ALP:
            rg["k"][r] = o["k"][r]
            c = o["combos"][r].copy()
            c[c < 0] = 0
            rg["combos"][r] = c.astype(np.uint8)
        else:
            rg["rd_rbw"][r] = o["bw"][v0]
            rg["rd_lbw"][r] = o["lbw"][v0]
            rg["rd_dict_size"][r] = o["dict_size"][r]
            rg["rd_dict"][r] = o["dict"][r]
    psz, esz = record_sizes(o["scheme"], o["bw"], o["lbw"], o["exc_cnt"], W)
    poff = np.concatenate([[0], np.cumsum(psz)]).astype(np.uint64)
    eoff = np.concatenate([[0], np.cumsum(esz)]).astype(np.uint64)
    vec = np.zeros(n, VECTOR_DTYPE)
    vec["packed_off"] = poff[:-1]
    vec["exc_off"] = eoff[:-1]
    vec["base"] = o["base"]
    vec["bw"] = o["bw"]
    vec["e"] = o["e"]
    vec["f"] = o["f"]
    vec["lbw"] = o["lbw"]
    vec["exc_cnt"] = o["exc_cnt"]
    vec["scheme"] = o["scheme"]
    packed = np.zeros(int(poff[-1]), np.uint8)
    exc = np.zeros(int(eoff[-1]), np.uint8)
    pk8 = o["packed"].view(np.uint8).reshape(n, 1024 * W)
    pl8 = o["packed_left"].view(np.uint8).reshape(n, 2048)
    ex8 = o["exc"].view(np.uint8).reshape(n, 1024 * W)
    ps8 = o["pos"].view(np.uint8).reshape(n, 2048)
    for v in range(n):
        b, c = int(o["bw"][v]), int(o["exc_cnt"][v])
        p0, e0 = int(poff[v]), int(eoff[v])
        packed[p0:p0 + 128 * b] = pk8[v, :128 * b]
        if o["scheme"][v] == SCHEME_ALP:
            exc[e0:e0 + W * c] = ex8[v, :W * c]
            exc[e0 + W * c:e0 + (W + 2) * c] = ps8[v, :2 * c]
        else:
            lb = int(o["lbw"][v])
            packed[p0 + 128 * b:p0 + 128 * (b + lb)] = pl8[v, :128 * lb]
            exc[e0:e0 + 2 * c] = ex8[v, :2 * c]
            exc[e0 + 2 * c:e0 + 4 * c] = ps8[v, :2 * c]
    return rg, vec, packed, exc


def expand(rg, vec, packed, exc, value_bytes=8):
    """inverse of compact(): -> dict with the fixed-stride arrays the checkers use (only the used prefix of each
    stride is filled)"""
    n = vec.size
    nrg = (n + 99) // 100
    W = value_bytes
    wdt, fdt = (np.int64, np.float64) if W == 8 else (np.int32, np.float32)
    o = dict(
        scheme=vec["scheme"].astype(np.uint8), e=vec["e"].copy(), f=vec["f"].copy(), bw=vec["bw"].copy(),
        lbw=vec["lbw"].copy(), base=vec["base"].copy(), exc_cnt=vec["exc_cnt"].copy(),
        packed=np.zeros((n, 1024), wdt), packed_left=np.zeros((n, 1024), np.uint16),
        exc=np.zeros((n, 1024), fdt), pos=np.zeros((n, 1024), np.uint16),
        dict=np.zeros((nrg, 8), np.uint16), dict_size=np.zeros(nrg, np.uint8), k=np.zeros(nrg, np.uint8),
        combos=np.full((nrg, 10), -1, np.int32))
    for r in range(nrg):
        if rg["scheme"][r] == SCHEME_ALP:
            o["k"][r] = rg["k"][r]
            o["combos"][r, :2 * int(rg["k"][r])] = rg["combos"][r, :2 * int(rg["k"][r])]
        else:
            o["dict"][r] = rg["rd_dict"][r]
            o["dict_size"][r] = rg["rd_dict_size"][r]
    pk8 = o["packed"].view(np.uint8).reshape(n, 1024 * W)
    pl8 = o["packed_left"].view(np.uint8).reshape(n, 2048)
    ex8 = o["exc"].view(np.uint8).reshape(n, 1024 * W)
    ps8 = o["pos"].view(np.uint8).reshape(n, 2048)
    for v in range(n):
        b, c = int(vec["bw"][v]), int(vec["exc_cnt"][v])
        p0, e0 = int(vec["packed_off"][v]), int(vec["exc_off"][v])
        pk8[v, :128 * b] = packed[p0:p0 + 128 * b]
        if vec["scheme"][v] == SCHEME_ALP:
            ex8[v, :W * c] = exc[e0:e0 + W * c]
            ps8[v, :2 * c] = exc[e0 + W * c:e0 + (W + 2) * c]
        else:
            lb = int(vec["lbw"][v])
            pl8[v, :128 * lb] = packed[p0 + 128 * b:p0 + 128 * (b + lb)]
            ex8[v, :2 * c] = exc[e0:e0 + 2 * c]
            ps8[v, :2 * c] = exc[e0 + 2 * c:e0 + 4 * c]
    return o
