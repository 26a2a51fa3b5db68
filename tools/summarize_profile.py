#!/usr/bin/env python3
"""Turns the outputs of tools/profile_round.sh (gpurun_out/prof_<tag>/) into the committed artefacts under profiles/:

  profiles/<tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary of `bench.py` (per-kernel calls / avg / total)
  profiles/<tag>_pmc_traffic.json   per-kernel HBM traffic per launch from the separate --pmc FETCH_SIZE / WRITE_SIZE
                                    passes: bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 -- on gfx950 FETCH_SIZE reports
                                    half the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM section)
  profiles/<tag>_bench.json         the bench line of the same run
  profiles/<tag>_launch_table.json  per-(kernel, layer) HIP-event durations + algorithmic flops/bytes (library profiler)
  profiles/<tag>_sq_counters.json   per kernel (and family): matrix-pipe busy share, VALU / MFMA instruction counts, wait
                                    shares, LDS bank conflicts from the two SQ --pmc passes (train steps only)
  profiles/<tag>_infer_kernel_stats.csv   the same --stats summary for the hipGraph-replayed batch-128 inference forward
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", f"prof_{tag}")
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)


def short(name):
    return name.replace("void oct::", "").replace("oct::", "").split("(")[0].replace(", ", ",")


stats = max(glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv")), key=os.path.getmtime)   # newest run
shutil.copy(stats, os.path.join(dst, f"{tag}_kernel_stats.csv"))
for f in ("bench.json", "launch_table.json"):
    shutil.copy(os.path.join(src, f), os.path.join(dst, f"{tag}_{f}"))

traffic = collections.defaultdict(lambda: dict(launches=0, fetch_kb=0.0, write_kb=0.0))
for counter, key in (("FETCH_SIZE", "fetch_kb"), ("WRITE_SIZE", "write_kb")):
    f = max(glob.glob(os.path.join(src, f"pmc_{counter}", "*", "*_counter_collection.csv")), key=os.path.getmtime)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter or "oct::" not in r["Kernel_Name"]:
            continue
        # only the per-rank-batch-32 training launches + batch-32 inference of this pass: all launches of the pass
        t = traffic[short(r["Kernel_Name"])]
        t[key] += float(r["Counter_Value"])
        if counter == "FETCH_SIZE":
            t["launches"] += 1
import subprocess
try:
    commit = subprocess.check_output(["git", "-C", root, "rev-parse", "--short", "HEAD"], text=True).strip()
except Exception:  # noqa: BLE001
    commit = "unknown"
try:
    pmc_steps = int(open(os.path.join(src, "pmc_steps.txt")).read().strip())
except (OSError, ValueError):
    pmc_steps = None
out = {}
for k, t in traffic.items():
    n = max(t["launches"], 1)
    out[k] = {"launches_in_pass": t["launches"],
              "fetch_size_kb_per_launch": t["fetch_kb"] / n, "write_size_kb_per_launch": t["write_kb"] / n,
              "hbm_bytes_per_launch": (2.0 * t["fetch_kb"] + t["write_kb"]) / n * 1024.0}
fam = collections.defaultdict(lambda: dict(launches=0, bytes=0.0))
total = 0.0
for k, v in out.items():
    f = fam[k.split("<")[0]]
    f["launches"] += v["launches_in_pass"]; f["bytes"] += v["hbm_bytes_per_launch"] * v["launches_in_pass"]
    total += v["hbm_bytes_per_launch"] * v["launches_in_pass"]
families = {k: {"launches_in_pass": v["launches"], "hbm_bytes_per_launch": v["bytes"] / max(v["launches"], 1),
                "hbm_bytes_per_step": (v["bytes"] / pmc_steps) if pmc_steps else None} for k, v in fam.items()}
doc = {"note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) of `bench.py --steps N --warmup 0 --no-inference` "
               "(train steps only; the pass also holds the one-off allocator / init kernels of torch, not counted: only oct:: "
               "kernels); hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 FETCH_SIZE correction, MI355X_MICROARCH.md)",
       "commit": commit, "train_steps_in_pass": pmc_steps, "kernels": out, "families": families}
if pmc_steps:
    doc["train_step_hbm_bytes"] = total / pmc_steps
json.dump(doc, open(os.path.join(dst, f"{tag}_pmc_traffic.json"), "w"), indent=1)
if pmc_steps:
    print(f"train-step HBM traffic: {total / pmc_steps / 1e9:.2f} GB/step over {pmc_steps} steps")

# ---- SQ counters: per kernel instantiation averages per launch, and what they say ----
sq = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for d in sorted(glob.glob(os.path.join(src, "sq_pass*"))):
    cand = glob.glob(os.path.join(d, "*", "*_counter_collection.csv"))
    if not cand:
        continue
    f = max(cand, key=os.path.getmtime)          # newest run of this pass only (older runs may still lie in the directory)
    for r in csv.DictReader(open(f)):
        if "oct::" not in r["Kernel_Name"]:
            continue
        a = sq[short(r["Kernel_Name"])][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
if sq:
    def derive(c):
        g = lambda k: c.get(k, 0.0)
        d = {}
        if g("SQ_BUSY_CYCLES"):      # SQ_VALU_MFMA_BUSY_CYCLES counts cycles, summed over the SIMDs that were busy; SQ_BUSY_CYCLES per SE
            d["mfma_busy_over_4x_sq_busy"] = round(g("SQ_VALU_MFMA_BUSY_CYCLES") / (4.0 * g("SQ_BUSY_CYCLES")), 4)
        if g("SQ_WAVE_CYCLES"):
            d["issue_stall_share_of_wave_cycles"] = round(g("SQ_WAIT_INST_ANY") / g("SQ_WAVE_CYCLES"), 4)
            d["parked_share_of_wave_cycles"] = round(g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES"), 4)
            d["issuing_share_of_wave_cycles"] = round(g("SQ_ACTIVE_INST_ANY") / g("SQ_WAVE_CYCLES"), 4)
        if g("SQ_INSTS_MFMA"):
            d["valu_per_mfma_instruction"] = round(g("SQ_INSTS_VALU") / g("SQ_INSTS_MFMA"), 2)
        if g("SQ_LDS_IDX_ACTIVE"):
            d["lds_conflict_share_of_lds_cycles"] = round(g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE"), 4)
        return d
    kern = {}
    famc = collections.defaultdict(lambda: collections.defaultdict(float)); famn = collections.defaultdict(int)
    for k, cs in sq.items():
        avg = {c: v[0] / max(v[1], 1) for c, v in cs.items()}
        n = max(v[1] for v in cs.values())
        kern[k] = {"launches_in_pass": n, "per_launch": {c: round(v, 1) for c, v in sorted(avg.items())}, "derived": derive(avg)}
        for c, v in cs.items():
            famc[k.split("<")[0]][c] += v[0]
        famn[k.split("<")[0]] += n
    fams = {k: {"launches_in_pass": famn[k], "totals": {c: round(v, 1) for c, v in sorted(cs.items())}, "derived": derive(cs)}
            for k, cs in famc.items()}
    json.dump({"note": "rocprofv3 --pmc (two passes of 8 SQ counters, counters + kernel trace only) over `bench.py --steps 1 --warmup 1 "
                       "--no-inference ...`: train steps at B = 32.  SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles "
                       "summed over waves, SQ_VALU_MFMA_BUSY_CYCLES cycles (MI355X_MICROARCH.md); `derived` are ratios of counters of "
                       "the SAME pass.  Profiled runs clock lower than un-profiled ones: ratios, not absolute times.",
               "commit": commit, "families": fams, "kernels": kern},
              open(os.path.join(dst, f"{tag}_sq_counters.json"), "w"), indent=1)
    print("SQ counters:", ", ".join(f"{k} mfma-busy {v['derived'].get('mfma_busy_over_4x_sq_busy')}" for k, v in fams.items()
                                    if k in ("conv_bx_k", "conv_dwbx_k", "conv_bt_k", "conv_dwbt_k")))
inf = glob.glob(os.path.join(src, "trace_infer", "*", "*_kernel_stats.csv"))
if inf:
    shutil.copy(max(inf, key=os.path.getmtime), os.path.join(dst, f"{tag}_infer_kernel_stats.csv"))

rows = list(csv.DictReader(open(stats)))
print(f"{'kernel':52s} {'calls':>6s} {'avg_us':>9s} {'total_ms':>9s} {'%':>6s} {'HBM MB/launch':>14s}")
for r in rows[:24]:
    k = short(r["Name"])
    mb = out.get(k, {}).get("hbm_bytes_per_launch", 0) / 1e6
    print(f"{k[:52]:52s} {r['Calls']:>6s} {float(r['AverageNs'])/1e3:9.1f} {float(r['TotalDurationNs'])/1e6:9.2f} "
          f"{float(r['Percentage']):6.1f} {mb:14.1f}")
