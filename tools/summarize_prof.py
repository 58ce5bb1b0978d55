#!/usr/bin/env python3
"""Turn rocprofv3 CSV output (gpurun_out/...) into the small summaries committed under profiles/.

    python tools/summarize_prof.py stats  <kernel_stats.csv> <out.csv>
    python tools/summarize_prof.py pmc    <fetch_counter_collection.csv> <write_counter_collection.csv> <out.csv>
"""
import collections
import csv
import re
import sys


def short(name: str) -> str:
    m = re.search(r"(\w+_kernel\w*)(<[^>]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else name[:60]


def stats(src, dst):
    rows = list(csv.DictReader(open(src)))
    with open(dst, "w") as f:
        f.write("kernel,calls,avg_us,total_ms,percent\n")
        for r in rows:
            if "mri::" not in r["Name"]:
                continue
            f.write(f"\"{short(r['Name'])}\",{r['Calls']},{float(r['AverageNs']) / 1e3:.1f},"
                    f"{float(r['TotalDurationNs']) / 1e6:.3f},{r['Percentage']}\n")


def pmc(fetch, write, dst):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in (fetch, write):
        for r in csv.DictReader(open(path)):
            if "mri::" in r["Kernel_Name"]:
                agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    with open(dst, "w") as f:
        f.write("# rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes; values in KB per launch.\n")
        f.write("# gfx950: FETCH_SIZE tallies 128-B requests at 64 B, i.e. reports 1/2 of coalesced streaming\n")
        f.write("# reads (MI355X_MICROARCH.md, HBM); hbm_bytes_corrected = 2*FETCH + WRITE (KB * 1024).\n")
        f.write("kernel,launches,fetch_kb,write_kb,hbm_mb_corrected\n")
        for k, c in sorted(agg.items(), key=lambda kv: -sum(kv[1].get("WRITE_SIZE", [0]))):
            fe = sum(c.get("FETCH_SIZE", [0])) / max(1, len(c.get("FETCH_SIZE", [])))
            wr = sum(c.get("WRITE_SIZE", [0])) / max(1, len(c.get("WRITE_SIZE", [])))
            n = max(len(c.get("FETCH_SIZE", [])), len(c.get("WRITE_SIZE", [])))
            f.write(f"\"{k}\",{n},{fe:.1f},{wr:.1f},{(2 * fe + wr) * 1024 / 1e6:.1f}\n")


if __name__ == "__main__":
    {"stats": stats, "pmc": pmc}[sys.argv[1]](*sys.argv[2:])
