#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REAL reference (oracle/_ref/libalp_ref.so, built in place from
/root/reference by oracle/Makefile).  Runs only in the build container; the fixtures it writes are data
(inputs + the reference's outputs + the known-answer numbers the reference's own unit test asserts) and
travel to the GPU box, where /root/reference does not exist.

Inputs (reference's own test data, read the way test/test_alp_sample.cpp:118-134 reads it — whitespace
separated tokens parsed as doubles, first 1024 values):
  data/samples/*.csv (30 real columns), data/generated/generated_doubles_bw{0..64}.csv, data/edge_case/edge_case.csv,
  data/double/test_0.csv;  data/1_rg_data_sample/*.bin (5 x 131072 doubles = 128 vectors = 2 rowgroups).
Known answers: (bit_width, exceptions_count) per column from data/include/double/alp_dataset.hpp,
  data/include/generated_columns.hpp, data/include/edge_case.hpp, data/include/double/double_dataset.hpp,
  asserted by the reference at test/test_alp_sample.cpp:178-179.
"""
import glob
import os
import re
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle.pyoracle import Reference, ReferenceF32  # noqa: E402
import datagen  # noqa: E402

REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")


def read_csv_first_vector(path):
    toks = open(path).read().split()
    vals = np.array([float(t) for t in toks[:1024]], dtype=np.float64)
    assert vals.size == 1024, (path, vals.size)
    return vals


def parse_descriptors(hpp):
    """{csv basename: (exceptions_count, bit_width, suitable_for_cutting)} from an ALPColumnDescriptor table.
    Field order (data/include/column.hpp:31-41): id, name, csv, bin, factor, exponent, exceptions_count, bit_width[, cutting]"""
    txt = open(hpp).read()
    out = {}
    for m in re.finditer(r'\{\s*(\d+)\s*,\s*"([^"]+)"\s*,([^{}]*?)\}', txt, re.S):
        body = m.group(3)
        csv = re.search(r'"([^"]*\.csv)"', body)
        if not csv:
            continue
        nums = re.findall(r'(?<![\w."])(\d+|true|false)(?![\w."])', re.sub(r'"[^"]*"', '', body))
        nums = [n for n in nums]
        # the four trailing numbers are factor, exponent, exceptions_count, bit_width (+ optional bool)
        cutting = False
        if nums and nums[-1] in ("true", "false"):
            cutting = nums[-1] == "true"
            nums = nums[:-1]
        fac, exp, exc, bw = (int(x) for x in nums[-4:])
        out[os.path.basename(csv.group(1))] = dict(name=m.group(2), exc=exc, bw=bw, cutting=cutting, fac=fac, exp=exp)
    return out


def main():
    R = Reference()
    os.makedirs(OUT, exist_ok=True)

    desc = {}
    for hpp in ("data/include/double/alp_dataset.hpp", "data/include/generated_columns.hpp", "data/include/edge_case.hpp",
                "data/include/double/double_dataset.hpp"):
        p = os.path.join(REF, hpp)
        if os.path.exists(p):
            desc.update(parse_descriptors(p))

    files = (sorted(glob.glob(f"{REF}/data/samples/*.csv")) +
             sorted(glob.glob(f"{REF}/data/generated/*.csv"), key=lambda p: int(re.findall(r"bw(\d+)", p)[0])) +
             [f"{REF}/data/edge_case/edge_case.csv", f"{REF}/data/double/test_0.csv"])
    names, inputs, outs, known = [], [], [], []
    for p in files:
        col = read_csv_first_vector(p)
        o = R.encode_column(col)
        names.append(os.path.relpath(p, f"{REF}/data"))
        inputs.append(col.view(np.uint64))
        outs.append(o)
        d = desc.get(os.path.basename(p))
        known.append((d["bw"], d["exc"], int(d["cutting"])) if d else (-1, -1, -1))
    stack = lambda k: np.concatenate([o[k] for o in outs], axis=0)
    np.savez_compressed(
        os.path.join(OUT, "first_vectors.npz"),
        names=np.array(names), input_bits=np.stack(inputs), known_bw_exc_cut=np.array(known, np.int32),
        scheme=stack("scheme"), e=stack("e"), f=stack("f"), bw=stack("bw"), lbw=stack("lbw"), base=stack("base"),
        exc_cnt=stack("exc_cnt"), packed=stack("packed"), packed_left=stack("packed_left"),
        exc_bits=stack("exc").view(np.uint64), pos=stack("pos"), dict=stack("dict"), dict_size=stack("dict_size"),
        k=stack("k"), combos=stack("combos"))
    print("first_vectors:", len(names), "columns;", sum(1 for k in known if k[0] >= 0), "with known answers")

    rg = {}
    for p in sorted(glob.glob(f"{REF}/data/1_rg_data_sample/*.bin")):
        col = np.fromfile(p, np.float64)
        o = R.encode_column(col)
        key = os.path.basename(p).replace(".bin", "")
        rg[key + "__input_bits"] = col.view(np.uint64)
        for k, v in o.items():
            rg[key + "__" + k] = v.view(np.uint64) if v.dtype == np.float64 else v
        print(key, "sum bw", int(o["bw"].sum()), "sum exc", int(o["exc_cnt"].sum()), "k", o["k"].tolist())
    np.savez_compressed(os.path.join(OUT, "rowgroup_samples.npz"), **rg)
    float_fixtures()
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


# the reference's float test columns and the answers its unit test asserts: data/include/float/test.hpp:10-14
# (bit widths), data/include/float/edge_case.hpp:10 (exceptions 192, bit width 0); read the way
# test/test_alp_sample.cpp:118-134 does (std::stof per token; avx512dq.csv has 1023 values, the rest of the buffer is 0)
FLOAT_KNOWN = {"float/test_0.csv": (4, -1), "float/test_1.csv": (10, -1), "float/test_2.csv": (17, -1),
               "float/test_3.csv": (0, -1), "edge_case/avx512dq.csv": (0, 192)}


def float_fixtures():
    R = ReferenceF32()
    z = {}
    names = []
    for rel, known in FLOAT_KNOWN.items():
        vals = [np.float32(t.rstrip(",")) for t in open(f"{REF}/data/{rel}").read().split()][:1024]
        col = np.array(vals + [0.0] * (1024 - len(vals)), np.float32)
        key = rel.replace("/", "_").replace(".csv", "")
        names.append(key)
        z[key + "__known_bw_exc"] = np.array(known, np.int32)
        z[key + "__input_bits"] = col.view(np.uint32)
        for k, v in R.encode_column(col).items():
            z[key + "__" + k] = v.view(np.uint32) if v.dtype == np.float32 else v
    synth = {"mixed_1pct": datagen.mixed_column_f32(120, seed=3, exc_rate=0.01),
             "rd_unit": datagen.rd_column_f32(110, seed=5, kind="unit"),
             "drifting_k": datagen.drifting_column_f32(110, seed=7),
             "adversarial": np.concatenate(list(datagen.adversarial_vectors_f32().values()))}
    for key, col in synth.items():
        names.append(key)
        z[key + "__known_bw_exc"] = np.array((-1, -1), np.int32)
        z[key + "__input_bits"] = col.view(np.uint32)
        o = R.encode_column(col)
        for k, v in o.items():
            z[key + "__" + k] = v.view(np.uint32) if v.dtype == np.float32 else v
        print("float", key, "schemes", np.unique(o["scheme"]).tolist(), "sum bw", int(o["bw"].sum()), "sum exc", int(o["exc_cnt"].sum()))
    z["names"] = np.array(names)
    np.savez_compressed(os.path.join(OUT, "float_vectors.npz"), **z)


if __name__ == "__main__":
    main()
