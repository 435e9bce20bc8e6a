#!/usr/bin/env python3
"""The per-kernel roofline table of DESIGN.md section 5 as markdown, from profiles/rNN_kernel_roofline.json:
python3 scripts/roofline_table.py profiles/r03_kernel_roofline.json"""
import json, sys

d = json.load(open(sys.argv[1]))
rows = sorted(d["rows"], key=lambda r: -r["share_of_native_time"])
print(f"| call (N = {d['nodes'] / 1e6:.2f} M rows) | per step | alg. GB | µs | TB/s | of 8 TB/s | share |")
print("|---|---|---|---|---|---|---|")
for r in rows:
    if r["share_of_native_time"] < 0.004:
        continue
    print(f"| {r['signature']} | {r['calls_per_step']} | {r['algorithmic_bytes'] / 1e9:.2f} | {r['avg_us']:.0f} | {r['GBps'] / 1e3:.2f} | "
          f"{r['frac_of_8TBps']:.2f} | {100 * r['share_of_native_time']:.1f} % |")
print(f"\nsum of native time per step: {d['sum_native_us_per_step']:.0f} µs")
