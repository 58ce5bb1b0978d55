#!/usr/bin/env python3
"""Print the interesting fields of bench.py JSON lines (tuning helper).

    python tools/benchline.py gpurun_out/b_cfg4.json [more.json ...]   # never reads stdin
"""
import json
import sys

for path in sys.argv[1:]:
    d = json.loads(open(path).read().strip().splitlines()[-1])
    print(path, round(d["ms_per_step"], 4), "%.4e" % d["value"], d["roofline"]["kernel"],
          round(d["roofline"]["frac"], 3), d["phases_ms"])
