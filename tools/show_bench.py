#!/usr/bin/env python3
"""Prints value / ms_per_step and selected per-kernel averages of a bench.py JSON line: show_bench.py file.json [substr ...]"""
import json
import sys

d = json.load(open(sys.argv[1]))
print(d["value"], d["ms_per_step"], "fwd_ms", d.get("fwd_ms"))
keys = sys.argv[2:]
print({k: round(v["avg_ms"], 4) for k, v in d.get("kernels", {}).items() if not keys or any(s in k for s in keys)})
