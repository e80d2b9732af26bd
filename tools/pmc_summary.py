"""Turn the rocprofv3 outputs of tools/profile_round.sh into the two small files kept under profiles/:
<tag>_kernel_stats.csv (copy of the --stats table) and <tag>_pmc_traffic.json (HBM bytes per launch of
the scoring kernel: FETCH_SIZE doubled per MI355X_MICROARCH.md, + WRITE_SIZE; counter unit KiB)."""
import csv
import glob
import json
import os
os.environ.setdefault("GAML_HIP_FLAVOUR", "dev")  # tools look inside the library: the development build
import shutil
import sys

sys.path.insert(0, os.getcwd())
from bench import source_hash  # the sources the profile was taken on: bench.py prints roofline.traffic only when it matches

tag = sys.argv[1]
wl = sys.argv[2] if len(sys.argv) > 2 else ""   # "cfg3x8": the passes of tools/profile_round.sh with --workload cfg3x8 (directories x8_*)
pre = "x8_" if wl == "cfg3x8" else ""
name = f"{tag}_{wl}" if wl else tag
root = f"gpurun_out/prof_{tag}"


def find(sub, pat):
    f = sorted(glob.glob(os.path.join(root, sub, "**", pat), recursive=True))
    return f[0] if f else None


stats = find(pre + "trace", "*kernel_stats.csv")
if stats:
    shutil.copy(stats, os.path.join(root, f"{name}_kernel_stats.csv"))
res = {}
for sub, counter in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    f = find(pre + sub, "*counter_collection.csv")
    if not f:
        continue
    rows = [r for r in csv.DictReader(open(f)) if "paired_score_kernel" in r.get("Kernel_Name", "") and r.get("Counter_Name") == counter]
    # one row per dispatch (and per dimension instance: sum the instances of a dispatch)
    per = {}
    for r in rows:
        per[r["Dispatch_Id"]] = per.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
    vals = [per[k] for k in sorted(per, key=int)][-24:]  # the timed steps
    if vals:
        res[counter + "_KB_mean"] = sum(vals) / len(vals)
        res["dispatches"] = len(vals)
if "FETCH_SIZE_KB_mean" in res and "WRITE_SIZE_KB_mean" in res:
    res["hbm_bytes_per_launch"] = (2 * res["FETCH_SIZE_KB_mean"] + res["WRITE_SIZE_KB_mean"]) * 1024
out = {
    "command": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes) --output-format csv -- python3 bench.py --no-cpu-baseline "
               "--no-extras --steps 24 --warmup 24 (" + (wl or "cfg3") + "; last 24 dispatches = timed steps)",
    "workload": wl or "cfg3",
    "source_hash": source_hash(),
    "correction": "FETCH_SIZE doubled (gfx950 reports half the bytes of wide coalesced reads, MI355X_MICROARCH.md HBM); counter unit KiB",
    "round": tag,
    "kernels": {"paired_score_kernel": res},
}
json.dump(out, open(os.path.join(root, f"{name}_pmc_traffic.json"), "w"), indent=1)
print(json.dumps(out))
