#!/usr/bin/env python3
"""Compact view of a bench.py line: value, value_inflight1 and the per-kernel table.  usage: kernel_table.py < line.json"""
import json, sys
for line in sys.stdin:
    if not line.startswith("{"):
        continue
    d = json.loads(line)
    print(d["config"]["workload"][:60], "| value", d["value"], "inflight1", d["value_inflight1"], "ms/step", d["ms_per_step"],
          "| roofline", d["roofline"]["kernel"], d["roofline"]["frac"])
    for k in d["kernels"]:
        print("   %-42s L%-2d overlapped %.4f ms  isolated %s ms  %7.1f GB/s" % (k["name"], k["level"], k["ms"], k["isolated_ms"], k["GBps"]))
