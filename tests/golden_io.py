"""Loaders for tests/golden/*.npz (written by tools/make_golden.py from the real reference)."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
KEYS = ("scheme", "e", "f", "bw", "lbw", "base", "exc_cnt", "packed", "packed_left", "pos", "dict", "dict_size", "k", "combos")


def first_vectors():
    z = np.load(os.path.join(GOLDEN, "first_vectors.npz"))
    n = z["names"].size
    cases = []
    for i in range(n):
        o = {k: z[k][i:i + 1] for k in KEYS}
        o["exc"] = z["exc_bits"][i:i + 1].view(np.float64)
        cases.append((str(z["names"][i]), z["input_bits"][i].view(np.float64), o, z["known_bw_exc_cut"][i]))
    return cases


def rowgroup_samples():
    z = np.load(os.path.join(GOLDEN, "rowgroup_samples.npz"))
    names = sorted({k.split("__")[0] for k in z.files})
    cases = []
    for nm in names:
        o = {k: z[f"{nm}__{k}"] for k in KEYS}
        o["exc"] = z[f"{nm}__exc"].view(np.float64)
        cases.append((nm, z[f"{nm}__input_bits"].view(np.float64), o))
    return cases


def float_vectors():
    """[(name, float32 column, reference outputs, (known bit width, known exception count) or -1)]"""
    z = np.load(os.path.join(GOLDEN, "float_vectors.npz"))
    cases = []
    for nm in [str(x) for x in z["names"]]:
        o = {k: z[f"{nm}__{k}"] for k in KEYS}
        o["exc"] = z[f"{nm}__exc"].view(np.float32)
        cases.append((nm, z[f"{nm}__input_bits"].view(np.float32), o, z[f"{nm}__known_bw_exc"]))
    return cases


def assert_same_encoding(a, b, what="", word=np.uint64):
    """bit-exact comparison of two fixed-stride encode outputs (only the used prefix of exception arrays)"""
    n = a["scheme"].size
    for k in ("scheme", "e", "f", "bw", "lbw", "base", "exc_cnt", "packed", "packed_left", "dict", "dict_size", "k", "combos"):
        assert np.array_equal(a[k], b[k]), f"{what}: field {k} differs"
    for v in range(n):
        c = int(a["exc_cnt"][v])
        assert np.array_equal(a["pos"][v, :c], b["pos"][v, :c]), f"{what}: exception positions differ in vector {v}"
        if a["scheme"][v] == 2:
            x, y = a["exc"][v].view(word)[:c], b["exc"][v].view(word)[:c]
        else:
            x, y = a["exc"][v].view(np.uint16)[:c], b["exc"][v].view(np.uint16)[:c]
        assert np.array_equal(x, y), f"{what}: exception values differ in vector {v}"
