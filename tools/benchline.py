#!/usr/bin/env python3
"""Print the interesting fields of a bench.py JSON line read from stdin (tuning helper)."""
import json
import sys

d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(sys.argv[1] if len(sys.argv) > 1 else "", round(d["ms_per_step"], 4), "%.4e" % d["value"],
      d["roofline"]["kernel"], round(d["roofline"]["frac"], 3), d["phases_ms"])
