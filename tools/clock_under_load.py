#!/usr/bin/env python3
"""clock_under_load.py [kind] [seconds]: runs alpgpu_encode_f64 (or the decode of the benchmark column: kind = decode) back to back for a few seconds
while a second thread samples the GPU's shader clock and socket power (rocm-smi): is a VALU-heavy kernel running at the 2.4 GHz the roofline
arithmetic assumes, or at what the power budget leaves?  (MI355X_MICROARCH.md "DVFS give-back": 1.9-2.3 GHz under load.)"""
import os, subprocess, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from alp_amd import capi

kind = sys.argv[1] if len(sys.argv) > 1 else "mixed"
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
n = 1 << 20
ctx = capi.Context(0)
dev = torch.device("cuda:0")
if kind == "decode":
    col, _, _ = bench.build_decode_column(n, 0, seed=42)
    out = torch.empty(n * 1024, dtype=torch.float64, device=dev)
    step = lambda: ctx.decode(col, out)
else:
    x = bench.synthetic_input(kind, n, dev, seed=42)
    col = capi.DeviceColumn(n, 0)
    step = lambda: ctx.encode(x, col)
for _ in range(3):
    step()
torch.cuda.synchronize()
samples, stop = [], False

def sampler():
    while not stop:
        try:
            o = subprocess.run(["rocm-smi", "-d", "0", "--showclocks", "--showpower", "--csv"], capture_output=True, text=True, timeout=5).stdout
            samples.append(o.strip().replace("\n", " | "))
        except Exception as e:
            samples.append(f"rocm-smi failed: {e}")
            break
        time.sleep(0.3)

th = threading.Thread(target=sampler)
th.start()
t0 = time.perf_counter(); k = 0
while time.perf_counter() - t0 < secs:
    for _ in range(20):
        step()
    torch.cuda.synchronize(); k += 20
el = time.perf_counter() - t0
stop = True; th.join()
print(f"{kind}: {k} launches in {el:.2f} s = {el / k * 1e3:.3f} ms each (kernel={os.environ.get('ALPGPU_ENCODE_KERNEL', 'default')})")
for s in samples[:3] + samples[len(samples) // 2: len(samples) // 2 + 3] + samples[-2:]:
    print("  ", s[:400])
