#!/usr/bin/env python3
"""What does the codec deliver when the column lives in HOST memory on both sides (the per-vector C++ header's situation, and a database
that compresses on the way to disk)?  A double-buffered pipeline over page-locked host buffers: chunk i is copied up on one stream while
chunk i-1 is encoded (rowgroup init + vector encode) and its compressed bytes — descriptors, packed stream, exception stream — are copied
down on another; then the same for decode (compressed up, doubles down).  The context follows torch's current stream, so the library's
work is enqueued on whichever stream the chunk uses.  Reports GB/s of doubles, PCIe included.   usage: pcie_pipeline.py [n_vectors] [chunk_vectors]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from alp_amd import capi
import bench

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512000
chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 12800  # whole rowgroups: 100 MiB of doubles per chunk
assert n % chunk == 0 and chunk % 100 == 0, "whole chunks of whole rowgroups"
dev = torch.device("cuda:0")
ctx = capi.Context(0)
x_dev = bench.synthetic_input("mixed", n, dev, seed=42)
host_in = torch.empty(n * 1024, dtype=torch.float64, pin_memory=True)
host_in.copy_(x_dev)
del x_dev
n_chunks = (n + chunk - 1) // chunk
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
bufs = [torch.empty(chunk * 1024, dtype=torch.float64, device=dev) for _ in range(2)]
cols = [capi.DeviceColumn(chunk, 0, packed_capacity=int(chunk * 8192 * 0.7) + 4096, exc_capacity=int(chunk * 8192 * 0.2) + 4096) for _ in range(2)]
# host side of the compressed column: generous page-locked slabs per chunk
h_vec = torch.empty(n_chunks, chunk * 32, dtype=torch.uint8, pin_memory=True)
h_rg = torch.empty(n_chunks, (chunk // 100 + 1) * 32, dtype=torch.uint8, pin_memory=True)
h_packed = torch.empty(n_chunks, int(chunk * 8192 * 0.7) + 4096, dtype=torch.uint8, pin_memory=True)
h_exc = torch.empty(n_chunks, int(chunk * 8192 * 0.2) + 4096, dtype=torch.uint8, pin_memory=True)
sizes = [None] * n_chunks


def encode_pass():
    for i in range(n_chunks):
        k = i & 1
        v0, v1 = i * chunk, min(n, (i + 1) * chunk)
        cnt = v1 - v0
        with torch.cuda.stream(streams[k]):
            if True:
                bufs[k][: cnt * 1024].copy_(host_in[v0 * 1024: v1 * 1024], non_blocking=True)
                ctx.encode(bufs[k][: cnt * 1024], cols[k])
                # the sizes are needed on the host to know how much to bring down: a small synchronous read (part of the price)
                pb, eb, ov = ctx.column_totals(cols[k])
                assert ov == 0
                sizes[i] = (cnt, pb, eb)
                h_vec[i, : cnt * 32].copy_(cols[k].vectors[: cnt * 32], non_blocking=True)
                nrg = (cnt + 99) // 100
                h_rg[i, : nrg * 32].copy_(cols[k].rowgroups[: nrg * 32], non_blocking=True)
                h_packed[i, :pb].copy_(cols[k].packed[:pb], non_blocking=True)
                h_exc[i, :eb].copy_(cols[k].exc[:eb], non_blocking=True)
    torch.cuda.synchronize()


host_out = torch.empty(n * 1024, dtype=torch.float64, pin_memory=True)


def decode_pass():
    for i in range(n_chunks):
        k = i & 1
        cnt, pb, eb = sizes[i]
        v0 = i * chunk
        with torch.cuda.stream(streams[k]):
            nrg = (cnt + 99) // 100
            cols[k].vectors[: cnt * 32].copy_(h_vec[i, : cnt * 32], non_blocking=True)
            cols[k].rowgroups[: nrg * 32].copy_(h_rg[i, : nrg * 32], non_blocking=True)
            cols[k].packed[:pb].copy_(h_packed[i, :pb], non_blocking=True)
            cols[k].exc[:eb].copy_(h_exc[i, :eb], non_blocking=True)
            ctx.decode(cols[k], bufs[k])
            host_out[v0 * 1024: (v0 + cnt) * 1024].copy_(bufs[k][: cnt * 1024], non_blocking=True)
    torch.cuda.synchronize()


def timed(f, reps=3):
    f()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        f()
        ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2]


te = timed(encode_pass)
td = timed(decode_pass)
ok = bool(torch.equal(host_out.view(torch.int64), host_in.view(torch.int64)))
comp = sum(s[1] + s[2] + 32 * s[0] for s in sizes)
gb = n * 8192 / 1e9
print(f"host-to-host, {n} vectors ({gb:.1f} GB of doubles) in chunks of {chunk}: encode {gb / te:.1f} GB/s (up {gb / te:.1f} + down {comp / 1e9 / te:.1f} GB/s on the link), "
      f"decode {gb / td:.1f} GB/s (up {comp / 1e9 / td:.1f} + down {gb / td:.1f} GB/s); round trip bit-exact: {ok}")
# the plain copies for scale: the same bytes with no codec in between
def copy_up():
    for i in range(n_chunks):
        k = i & 1
        v0, v1 = i * chunk, min(n, (i + 1) * chunk)
        with torch.cuda.stream(streams[k]):
            bufs[k][: (v1 - v0) * 1024].copy_(host_in[v0 * 1024: v1 * 1024], non_blocking=True)
    torch.cuda.synchronize()
def copy_down():
    for i in range(n_chunks):
        k = i & 1
        v0, v1 = i * chunk, min(n, (i + 1) * chunk)
        with torch.cuda.stream(streams[k]):
            host_out[v0 * 1024: v1 * 1024].copy_(bufs[k][: (v1 - v0) * 1024], non_blocking=True)
    torch.cuda.synchronize()
print(f"plain page-locked copies of the doubles: up {gb / timed(copy_up):.1f} GB/s, down {gb / timed(copy_down):.1f} GB/s")

# the library's own pipeline (include/alpgpu.h: alpgpu_compress_host_f64 / alpgpu_decompress_host_f64) on the same column
nv = n * 1024
cap = int(capi.lib.alpgpu_blob_size(n, capi.lib.alpgpu_packed_capacity(n), capi.lib.alpgpu_exc_capacity(n)))
blob_buf = torch.empty(cap, dtype=torch.uint8, pin_memory=True)
blob = ctx.compress_host(host_in, blob_buf)
tc = timed(lambda: ctx.compress_host(host_in, blob_buf))
td2 = timed(lambda: ctx.decompress_host(blob, host_out))
ok2 = bool(torch.equal(host_out.view(torch.int64), host_in.view(torch.int64)))
print(f"alpgpu_compress_host_f64: {gb / tc:.1f} GB/s of doubles into a {blob.numel() / 1e9:.2f} GB blob; alpgpu_decompress_host_f64: {gb / td2:.1f} GB/s; round trip bit-exact: {ok2}")
pageable = host_in.clone()
tcp = timed(lambda: ctx.compress_host(pageable, torch.empty(cap, dtype=torch.uint8)), reps=1)
print(f"the same from and to pageable memory: compress {gb / tcp:.1f} GB/s")
