#!/usr/bin/env python3
"""Prints the region stamps of the wave-group kernel (a -DFSMC_REGION_STAMPS build; fsmc_kernels_w2.h) from a bench.py
JSON line: per wave role, the share of its cycles spent in every region.  Usage: tools/decode_w2_stamps.py <bench.json>"""
import json
import sys

NAMES = {0: "b.ph0 work", 1: "b.ph1 work", 2: "b.ph2 work", 3: "b.ph3 work", 4: "b.ph0 barrier", 5: "b.ph1 barrier",
         6: "b.ph2 barrier", 7: "b.ph3 barrier", 8: "b.sum", 9: "b.scale", 10: "a.ph0 work", 11: "a.ph1 work",
         12: "a.ph2 work", 13: "a.ph3 work", 14: "a.ph0 barrier", 15: "a.ph1 barrier", 16: "a.ph2 barrier",
         17: "a.ph3 barrier", 18: "a.sum", 19: "a.scale", 20: "between b steps", 21: "in front of a step",
         22: "combine", 23: "combine sum", 24: "consumer rest", 25: "next-row requests", 26: "scan sum",
         27: "scan decision", 28: "scan barrier", 29: "b loop: wait for emission rows"}
d = json.load(open(sys.argv[1]))
pc = d["config"]["phase_cycles"]
tot = [sum(pc[8 + 30 * h:8 + 30 * h + 30]) for h in range(4)]
print("kernel_ms", d["roofline"]["kernel_ms"])
print("%-20s" % "region" + "".join("%9s" % ("wave%d" % h) for h in range(4)))
for r in range(30):
    if any(pc[8 + 30 * h + r] for h in range(4)):
        print("%-20s" % NAMES[r] + "".join("%8.1f%%" % (100 * pc[8 + 30 * h + r] / tot[h]) for h in range(4)))
