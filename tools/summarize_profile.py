"""Turns gpurun_out/prof (tools/profile_bench.sh) into the committed summaries:
profiles/<tag>_rocprofv3.json, profiles/<tag>_kernel_stats.csv and profiles/pmc_traffic.json
(the HBM bytes per launch bench.py reports as roofline.traffic)."""
import collections, csv, glob, json, os, shutil, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
prof = os.path.join(root, "gpurun_out", "prof")
out = {"command": "python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline (cfg2: 1200x675x100spp, 1 GPU)"}
ks = glob.glob(os.path.join(prof, "trace", "*", "*_kernel_stats.csv"))[0]
shutil.copy(ks, os.path.join(root, "profiles", f"{tag}_kernel_stats.csv"))
rows = list(csv.DictReader(open(ks)))
out["kernel_stats"] = [{k: r[k] for k in ("Name", "Calls", "AverageNs", "Percentage")} for r in rows[:5]]
pm = {}
for f in sorted(glob.glob(os.path.join(prof, "pmc*", "*", "*_counter_collection.csv"))):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "render_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            out["vgpr_count"], out["lds_block_size"] = r["VGPR_Count"], r["LDS_Block_Size"]
    for k, v in agg.items():
        pm[k] = sum(v) / len(v)
out["pmc_mean_per_launch_render_kernel"] = pm
# HBM traffic per launch, corrected as MI355X_MICROARCH.md prescribes: FETCH_SIZE/WRITE_SIZE are in
# KiB; FETCH_SIZE reads exactly half the bytes of a wide coalesced stream on gfx950 (x2); WRITE_SIZE
# is exact for atomics and 16-B stores.
fetch = pm.get("FETCH_SIZE", 0.0) * 1024 * 2
write = pm.get("WRITE_SIZE", 0.0) * 1024
out["hbm_bytes_per_launch"] = {"fetch_corrected": fetch, "write": write, "total": fetch + write}
if "GRBM_GUI_ACTIVE" in pm:
    avg_ns = float(out["kernel_stats"][0]["AverageNs"])
    out["effective_clock_GHz"] = pm["GRBM_GUI_ACTIVE"] / 8 / avg_ns
json.dump(out, open(os.path.join(root, "profiles", f"{tag}_rocprofv3.json"), "w"), indent=1)
json.dump({"config": [1200, 675, 100, 1], "hbm_bytes_per_launch": fetch + write,
           "mfma_insts_per_launch": pm.get("SQ_INSTS_MFMA"), "valu_insts_per_launch": pm.get("SQ_INSTS_VALU"),
           "valu_busy": (pm["SQ_ACTIVE_INST_VALU"] * 4.0 / (pm["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
                         if "SQ_ACTIVE_INST_VALU" in pm and "GRBM_GUI_ACTIVE" in pm else None),
           "source": f"profiles/{tag}_rocprofv3.json (FETCH_SIZE x2 + WRITE_SIZE, separate --pmc passes)"},
          open(os.path.join(root, "profiles", "pmc_traffic.json"), "w"), indent=1)
print(json.dumps(out["kernel_stats"][0]), out["hbm_bytes_per_launch"], out.get("effective_clock_GHz"))
