"""Turns gpurun_out/prof_<cfg> (tools/profile_bench.sh) into the committed summaries:
profiles/<tag>_rocprofv3_<cfg>.json, profiles/<tag>_kernel_stats_<cfg>.csv and profiles/pmc_replay.json
(the counters bench.py replays into its `roofline` object, labelled with this source and with the sha of the
kernel sources they were measured on -- bench.py nulls them when the sources have changed)."""
import collections, csv, glob, json, os, shutil, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import bench  # noqa: E402  (kernel_source_sha)

CONFIGS = {"target": [1200, 675, 500, 1], "cfg2": [1200, 675, 100, 1], "tenk": [1920, 1080, 256, 1], "weak": [3840, 2160, 500, 1]}
entries = []
for cfg, shape in CONFIGS.items():
    prof = os.path.join(root, "gpurun_out", f"prof_{cfg}")
    if not os.path.isdir(prof):
        continue
    out = {"command": "python3 " + open(os.path.join(prof, "command.txt")).read().strip(), "config": shape,
           "kernel_source_sha": bench.kernel_source_sha()}
    newest = lambda pattern: max(glob.glob(pattern), key=os.path.getmtime)     # gpurun MERGES runs into gpurun_out/: take the latest
    ks = newest(os.path.join(prof, "trace", "*", "*_kernel_stats.csv"))
    shutil.copy(ks, os.path.join(root, "profiles", f"{tag}_kernel_stats_{cfg}.csv"))
    rows = list(csv.DictReader(open(ks)))
    out["kernel_stats"] = [{k: r[k] for k in ("Name", "Calls", "AverageNs", "MinNs", "MaxNs", "Percentage") if k in r} for r in rows[:5]]
    top_kernel = rows[0]["Name"].split("(")[0]            # only the launches of the timed configuration's kernel enter the means
    pm = {}
    for f in [newest(os.path.join(d, "*", "*_counter_collection.csv")) for d in sorted(glob.glob(os.path.join(prof, "pmc[0-9]*"))) if os.path.isdir(d)]:
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if r["Kernel_Name"].split("(")[0] == top_kernel:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
                out["vgpr_count"], out["lds_block_size"] = r["VGPR_Count"], r["LDS_Block_Size"]
        for k, v in agg.items():
            pm[k] = sum(v) / len(v)
    out["pmc_mean_per_launch_render_kernel"] = pm
    # HBM traffic per launch, corrected as MI355X_MICROARCH.md prescribes: FETCH_SIZE/WRITE_SIZE are in KiB;
    # FETCH_SIZE reads exactly half the bytes of a wide coalesced stream on gfx950 (x2); WRITE_SIZE is exact
    # for atomics and 16-B stores.
    fetch = pm.get("FETCH_SIZE", 0.0) * 1024 * 2
    write = pm.get("WRITE_SIZE", 0.0) * 1024
    out["hbm_bytes_per_launch"] = {"fetch_corrected": fetch, "write": write, "total": fetch + write}
    avg_ns = float(out["kernel_stats"][0]["AverageNs"])
    cycles = None
    if "GRBM_GUI_ACTIVE" in pm:
        cycles = pm["GRBM_GUI_ACTIVE"] / 8.0                      # summed over the 8 XCDs
        out["effective_clock_GHz"] = cycles / avg_ns
    valu_busy = pm["SQ_ACTIVE_INST_VALU"] * 4.0 / (cycles * 1024.0) if cycles and "SQ_ACTIVE_INST_VALU" in pm else None
    out["valu_busy"] = valu_busy
    json.dump(out, open(os.path.join(root, "profiles", f"{tag}_rocprofv3_{cfg}.json"), "w"), indent=1)
    entries.append({"config": shape, "kernel_source_sha": out["kernel_source_sha"],
                    "source": f"profiles/{tag}_rocprofv3_{cfg}.json: rocprofv3 --pmc passes of `{out['command']}` "
                              f"(separate runs; FETCH_SIZE x2 + WRITE_SIZE); NOT measured in this bench run",
                    "rocprof_avg_kernel_ms": avg_ns / 1e6,
                    "hbm_bytes_per_launch": fetch + write,
                    "mfma_insts_per_launch": pm.get("SQ_INSTS_MFMA"), "valu_insts_per_launch": pm.get("SQ_INSTS_VALU"),
                    "simd_cycles_per_launch": cycles, "valu_busy": valu_busy,
                    # instruction classes for the hardware-derived issue peak (bench.py, roofline.peak) and the overhead accounting
                    "f64_addmulfma_insts_per_launch": (sum(pm.get(k, 0.0) for k in ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64"))
                                                       if "SQ_INSTS_VALU_FMA_F64" in pm else None),
                    "trans_f32_insts_per_launch": pm.get("SQ_INSTS_VALU_TRANS_F32"),
                    "salu_insts_per_launch": pm.get("SQ_INSTS_SALU"), "branch_insts_per_launch": pm.get("SQ_INSTS_BRANCH"),
                    "lds_bank_conflict_share": (pm["SQ_LDS_BANK_CONFLICT"] / pm["SQ_LDS_IDX_ACTIVE"]
                                                if pm.get("SQ_LDS_IDX_ACTIVE") else None)})
    print(cfg, json.dumps(out["kernel_stats"][0]), out["hbm_bytes_per_launch"], out.get("effective_clock_GHz"), valu_busy)
json.dump({"entries": entries}, open(os.path.join(root, "profiles", "pmc_replay.json"), "w"), indent=1)
