#!/usr/bin/env python3
"""Copy what tools/gpu_final.sh left under gpurun_out/<round>/final (and the rocprofv3 counter CSVs under
gpurun_out/<round>/pmc_*) into the committed summaries under profiles/ and refresh profiles/traffic.json.

    python tools/collect_final.py [round-tag, default r03] [round-dir, default r3]
"""
import ast
import csv
import glob
import json
import os
import re
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
rdir = sys.argv[2] if len(sys.argv) > 2 else "r4"
src = os.path.join(ROOT, "gpurun_out", rdir, "final")
dst = os.path.join(ROOT, "profiles")


def last_line(path):
    return open(path).read().strip().splitlines()[-1]


for w in ("cfg2", "cfg3", "cfg4", "cfg5"):
    open(os.path.join(dst, f"{tag}_bench_{w}.json"), "w").write(last_line(os.path.join(src, f"bench_{w}.json")) + "\n")
for w in ("cfg3", "cfg4", "cfg5"):
    open(os.path.join(dst, f"{tag}_predict_{w}.json"), "w").write(last_line(os.path.join(src, f"predict_{w}.json")) + "\n")
for w in ("cfg3", "cfg4", "cfg4_packed", "cfg5", "cfg4_predict"):
    shutil.copy(os.path.join(src, f"{w}_kernel_stats.csv"), os.path.join(dst, f"{tag}_{w}_kernel_stats.csv"))

traffic = {}
for w in ("cfg3", "cfg4", "cfg4_packed", "cfg2", "cfg5"):
    if not glob.glob(os.path.join(ROOT, "gpurun_out", rdir, f"pmc_{w}_fetch", "*counter_collection.csv")):
        continue
    f = glob.glob(os.path.join(ROOT, "gpurun_out", rdir, f"pmc_{w}_fetch", "*counter_collection.csv"))[0]
    g = glob.glob(os.path.join(ROOT, "gpurun_out", rdir, f"pmc_{w}_write", "*counter_collection.csv"))[0]
    out = os.path.join(dst, f"{tag}_{w}_pmc_traffic.csv")
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "summarize_prof.py"), "pmc", f, g, out])
    mb = {}
    for r in csv.DictReader(l for l in open(out) if not l.startswith("#")):
        # one entry per kernel NAME: the template instances of a name (bin_kernel<.., true> the scatter,
        # bin_kernel<.., false> the count) are ADDED -- round 2 overwrote one with the other
        key = re.sub(r"<.*", "", r["kernel"])
        mb[key] = mb.get(key, 0.0) + float(r["hbm_mb_corrected"])
    traffic[w] = mb


def parse_log(path):
    rows = []
    for line in open(path):
        m = re.match(r"(\w+) (\{.*\}) launches (\d+)", line.strip())
        if m:
            rows.append((m.group(1), int(m.group(3)), ast.literal_eval(m.group(2))))
    return rows


cols = ["SQ_BUSY_CU_CYCLES", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY",
        "SQ_WAVE_CYCLES"]
for w in ("cfg4", "cfg3"):
    with open(os.path.join(dst, f"{tag}_{w}_pmc_sq.csv"), "w") as f:
        f.write("# rocprofv3 --pmc (one pass, tools/gpu_pmc.sh); per-launch averages, summed over the chip.\n")
        f.write("# mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES)\n")
        f.write("kernel,launches," + ",".join(cols) + ",mfma_busy\n")
        for k, n, d in parse_log(os.path.join(src, f"pmc_{w}_sq.log")):
            b = d.get("SQ_BUSY_CU_CYCLES", 0)
            f.write(f"\"{k}\",{n}," + ",".join(str(d.get(c, 0)) for c in cols) +
                    f",{(d.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (4 * b) if b else 0):.3f}\n")
with open(os.path.join(dst, f"{tag}_cfg4_pmc_lds.csv"), "w") as f:
    f.write("# rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES (tools/gpu_pmc.sh); "
            "per-launch averages, summed over the chip.\n# lds_busy = SQ_LDS_IDX_ACTIVE / SQ_BUSY_CU_CYCLES\n")
    f.write("kernel,launches,SQ_BUSY_CU_CYCLES,SQ_LDS_IDX_ACTIVE,SQ_LDS_BANK_CONFLICT,SQ_INSTS_MFMA,lds_busy\n")
    for k, n, d in parse_log(os.path.join(src, "pmc_cfg4_lds.log")):
        b = d.get("SQ_BUSY_CU_CYCLES", 0)
        f.write(f"\"{k}\",{n},{b},{d.get('SQ_LDS_IDX_ACTIVE', 0)},{d.get('SQ_LDS_BANK_CONFLICT', 0)},"
                f"{d.get('SQ_INSTS_MFMA', 0)},{(d.get('SQ_LDS_IDX_ACTIVE', 0) / b if b else 0):.3f}\n")

t3 = traffic["cfg3"]
tj = os.path.join(dst, "traffic.json")
d = json.load(open(tj))
d["_comment"] = (f"HBM bytes per launch of each bench.py phase, from profiles/{tag}_cfg{{3,4,4_packed}}_pmc_traffic.csv "
                 "(rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, 2*FETCH + WRITE as MI355X_MICROARCH.md "
                 "prescribes for gfx950). cfg4 = f32 table-gradient records (the headline), cfg4_packed = "
                 "bwd_records 1. Read by bench.py for roofline.traffic.")
for name in ("cfg4", "cfg4_packed", "cfg2", "cfg5"):
    if name not in traffic:
        continue
    t4 = traffic[name]
    decoder = next(v for k, v in t4.items() if k.startswith("tiny_mlp"))
    bwd4 = sum(v for k, v in t4.items() if k.startswith(("bin_kernel", "dense_and_accumulate", "bin_finalize",
                                                          "bin_chunk_scan", "bin_prefix", "bin_accumulate",
                                                          "dense_absmax", "dense_level")))
    d[name] = {"mlp_fused": int((decoder + t4.get("slab_reduce_kernel", 0)) * 1e6), "hashgrid_bwd": int(bwd4 * 1e6),
               "hashgrid_fwd": int(next(v for k, v in t4.items() if k.startswith("hashgrid_fwd")) * 1e6),
               "adam": int(t4["adam_kernel"] * 1e6)}
split = t3.get("siren_split_weights_kernel", 0) / 2  # launched by the forward and by the backward entry
fwd3 = next(v for k, v in t3.items() if k.startswith("siren_forward"))    # siren_forward_rows_kernel (H = 256) / siren_forward_kernel
bwd3 = next(v for k, v in t3.items() if k.startswith("siren_backward"))
d["cfg3"] = {"mlp_fwd": int((fwd3 + t3.get("siren_fwd_reduce_kernel", 0) + split) * 1e6),
             "mlp_bwd": int((bwd3 + t3.get("siren_bwd_reduce_kernel", 0) + split +
                             4 * (t3["siren_wgrad_kernel"] + t3.get("slab_sum_kernel", 0))) * 1e6),
             "adam": int(t3["adam_kernel"] * 1e6)}
json.dump(d, open(tj, "w"), indent=2)
for w in ("cfg4", "cfg2", "cfg3", "cfg5"):
    b = json.loads(last_line(os.path.join(src, f"bench_{w}.json")))
    print(w, round(b["ms_per_step"], 4), "%.3e" % b["value"], b["phases_ms"], "cpu %.3g" % b["cpu_baseline"]["value"],
          b["roofline"]["kernel"], round(b["roofline"]["frac"], 3), b.get("psnr"))
for w in ("cfg4", "cfg3", "cfg5"):
    b = json.loads(last_line(os.path.join(src, f"predict_{w}.json")))
    print("predict", w, round(b["ms_per_step"], 4), "%.3e" % b["value"], b["phases_ms"], round(b["roofline"]["frac"], 3))
