#!/usr/bin/env python3
"""design_table.py: DESIGN.md section 6.1's table rows from the committed bench lines (profiles/r04_bench.json, r04_bench_configs4_n1.json,
r04_kernel_stats.csv) — rewrites the rows between the table's header and the CPU row in place"""
import json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = json.load(open(os.path.join(ROOT, "profiles", "r04_bench.json")))
d4 = json.load(open(os.path.join(ROOT, "profiles", "r04_bench_configs4_n1.json")))
e, r, fp = d["extras"], d["roofline"], d["extras"]["float_path"]
stats = open(os.path.join(ROOT, "profiles", "r04_kernel_stats.csv")).read().split("\n")
avg = next(float(l.split('",')[-1].split(",")[2]) / 1e6 for l in stats if "bench.py" in l and "k_decode_column<1" in l)
sw, swe = e["decode_sweep_by_bit_width"]["summary"], e["decode_sweep_by_bit_width_2pct_exceptions"]["summary"]
em, er, e0, e10 = (e[k] for k in ("encode_alp_mixed", "encode_alp_rd", "encode_alp_mixed_exc0", "encode_alp_mixed_exc10"))
f = lambda x, n=3: f"{x:.{n}f}"
rows = f"""| **decode, configs[1]** (1 Mi vectors, widths 1–53 by rowgroup, no exceptions) | **{d['value']:.0f} GB/s decoded** ({r['achieved']:.0f} GB/s algorithmic), kernel {f(r['kernel_ms'])} ms | **{f(r['frac'])}** (0.776–0.808 on this round's boxes) | rocprofv3 average of `k_decode_column` {f(avg)} ms; PMC traffic {r['traffic'] / 1e9:.3f} GB against {r['algorithmic_bytes_per_launch'] / 1e9:.3f} GB algorithmic (+{(r['traffic'] / r['algorithmic_bytes_per_launch'] - 1) * 100:.1f} %) |
| decode, configs[4] at N = 1 (92.5 GB shard, `--gpus 1 --column-gb 100`) | {d4['value']:.0f} GB/s | {f(d4['roofline']['frac'], 4)} | `r04_bench_configs4_n1.json` (per-rank record inside) |
| decode sweep, EVERY width 1–53, no exceptions | min **{f(sw['min'])}** ({sw['argmin_bit_width']} bits), p10 {f(sw['p10'])}, mean **{f(sw['mean'])}**, max {f(sw['max'])} | | `r04_decode_floor.txt`; first closing profile of the round (before the residency rule): 0.712 / 0.724 / 0.756 |
| the same with 20 exceptions per vector (2 %) | min **{f(swe['min'])}** ({swe['argmin_bit_width']} bit), p10 {f(swe['p10'])}, mean {f(swe['mean'])} | | (first closing profile: 0.568 / 0.635 / 0.736) |
| measured ceilings on the same box | `copy_` {f(e['measured_copy_ceiling']['frac_of_nominal_peak'])}, `fill_` {f(e['measured_fill_ceiling']['frac_of_nominal_peak'])} | | |
| **encode, mixed decimal column** (search beside) | {em['input_GBps']:.0f} GB/s of input, **{f(em['ms'])} ms** | **{f(em['roofline_frac_algorithmic'])}** (round 3: 0.51–0.53; A/B boxes of this round: 2.94–3.02 ms = 0.544–0.558) | `k_encode_lean` ‖ `k_rowgroup_init<…8, true>` |
| encode, no exceptions / 10 % exceptions | {f(e0['ms'])} / {f(e10['ms'])} ms | {f(e0['roofline_frac_algorithmic'])} / {f(e10['roofline_frac_algorithmic'])} | |
| encode, all-ALP_RD column | {er['input_GBps']:.0f} GB/s, {f(er['ms'])} ms | **{f(er['roofline_frac_algorithmic'])}** (0.469–0.479 on this round's boxes; round 3: 0.450; first closing profile of this round 0.441) | |
| encode, configs[4] at N = 1 | {d4['encode']['value']:.0f} GB/s | {f(d4['encode']['roofline']['frac'])} (round 3: 0.538) | |
| decode → SUM, double (`k_sink_direct`) | {f(e['decode_sum_fused']['ms'])} ms | {f(e['decode_sum_fused']['roofline_frac_algorithmic'])} (0.62–0.66 on this round's boxes) | |
| float decode (decimal_mixed / rd) | | {f(fp['decimal_mixed']['decode_roofline_frac_algorithmic'])} / {f(fp['rd']['decode_roofline_frac_algorithmic'])} | `k_decode_column_f32<2>` |
| float encode | {f(fp['decimal_mixed']['encode_ms'], 2)} / {f(fp['rd']['encode_ms'], 2)} ms | {f(fp['decimal_mixed']['encode_roofline_frac_algorithmic'])} / {f(fp['rd']['encode_roofline_frac_algorithmic'])} (round 3: 0.29 / 0.31) | `r04_float_encode.txt` |
| float decode → SUM | {f(fp['decimal_mixed']['decode_sum_fused_ms'])} / {f(fp['rd']['decode_sum_fused_ms'])} ms | {f(fp['decimal_mixed']['decode_sum_roofline_frac_algorithmic'])} / {f(fp['rd']['decode_sum_roofline_frac_algorithmic'])} (round 3: 0.42) | |
"""
c = d["cpu_baseline"]
cpu = (f"decode {c['value']:.1f} GB/s on {c['cores']} threads ({c['single_thread_value']:.1f} one thread; the host is capped: {c['value'] / c['single_thread_value']:.1f} × one thread — "
       f"extrapolation {c['single_thread_value']:.1f} × 128 physical cores = {c['single_thread_value'] * 128 / 1000:.1f} TB/s, an upper bound, in the line); "
       f"encode {c['encode']['value']:.1f} GB/s ({c['encode']['single_thread_value']:.1f} one thread) over a 4 GiB sample")
p = os.path.join(ROOT, "DESIGN.md")
s = open(p).read()
a, b = s.index("| **decode, configs[1]** (1 Mi vectors"), s.index("| CPU, the real reference on the box's host cores")
s = s[:a] + rows + s[b:]
s = re.sub(r"decode [0-9.]+ GB/s on \d+ threads \([0-9.]+ one thread; the host is capped:.*?over a 4 GiB sample", cpu, s, count=1, flags=re.S)
open(p, "w").write(s)
print(rows)
