"""Per-kernel means of the rocprofv3 PMC passes written by tools/profile.sh, with derived figures.
usage: python tools/pmc_summary.py gpurun_out/prof_<tag> [--json out.json]"""
import collections, csv, glob, json, sys

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import kernel_source_hash  # noqa: E402

src = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
meta = {}
for f in glob.glob(f"{src}/pmc_*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "rtk::" not in name:
            continue
        agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
        meta[name] = {"VGPR": r["VGPR_Count"], "LDS": r["LDS_Block_Size"], "scratch": r["Scratch_Size"], "grid": r["Grid_Size"]}
times = {}
for f in glob.glob(f"{src}/trace/*/*_kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        name = r["Name"].split("(")[0].replace("void ", "")
        if "rtk::" in name:
            times[name] = float(r["AverageNs"]) / 1e6
out = {}
for name, counters in sorted(agg.items()):
    m = {k: sum(v) / len(v) for k, v in counters.items()}
    d = dict(meta[name])
    d["avg_ms"] = times.get(name)
    d.update(m)
    if "GRBM_GUI_ACTIVE" in m and m["GRBM_GUI_ACTIVE"] > 0:
        cyc = m["GRBM_GUI_ACTIVE"] / 8.0          # summed over the 8 XCDs
        d["cycles"] = cyc
        if d["avg_ms"]:
            d["clock_GHz"] = cyc / (d["avg_ms"] * 1e6)
        simd_cycles = 1024.0 * cyc
        if "SQ_INSTS_VALU" in m:
            d["valu_issue_util"] = m["SQ_INSTS_VALU"] * 2.0 / simd_cycles   # as if every wave64 VALU instruction took 2 cycles (FP32 rate)
        if "SQ_ACTIVE_INST_VALU" in m:
            # SQ_ACTIVE_INST_* count quad-cycles: time the SIMDs' vector ALUs are executing, whatever the
            # instruction costs (FP32 2 cycles, f64 4, f64 rcp/sqrt 16 per wave64)
            d["valu_busy"] = m["SQ_ACTIVE_INST_VALU"] * 4.0 / simd_cycles
            if "SQ_THREAD_CYCLES_VALU" in m and m["SQ_ACTIVE_INST_VALU"] > 0:
                d["valu_lane_util"] = m["SQ_THREAD_CYCLES_VALU"] / (m["SQ_ACTIVE_INST_VALU"] * 64.0)
        if "SQ_WAVE_CYCLES" in m:
            d["avg_waves_per_simd"] = m["SQ_WAVE_CYCLES"] * 4.0 / simd_cycles
        if "SQ_LDS_IDX_ACTIVE" in m:
            d["lds_busy"] = m["SQ_LDS_IDX_ACTIVE"] / (256.0 * cyc)
    if "SQ_INSTS_VALU_FMA_F64" in m and "GRBM_GUI_ACTIVE" in m and m["GRBM_GUI_ACTIVE"] > 0:
        # Instruction mix -> time the vector ALUs are executing.  Per wave64 instruction: f64
        # add/mul/fma 4 cycles, f64 transcendental (rcp, rsq, sqrt) 16, everything else
        # (FP32, integer, conversions, moves, compares) 2.
        f64 = m.get("SQ_INSTS_VALU_ADD_F64", 0) + m.get("SQ_INSTS_VALU_MUL_F64", 0) + m["SQ_INSTS_VALU_FMA_F64"]
        t64 = m.get("SQ_INSTS_VALU_TRANS_F64", 0)
        t32 = m.get("SQ_INSTS_VALU_TRANS_F32", 0)
        rest = m["SQ_INSTS_VALU"] - f64 - t64 - t32
        d["f64_share_of_valu_insts"] = (f64 + t64) / max(m["SQ_INSTS_VALU"], 1.0)  # a kernel that returns at once issues none
        if d.get("avg_ms") and d.get("valu_lane_util"):
            # flops actually executed: wave64 instructions x 64 lanes x share of lanes active (fma = 2)
            lanes = 64.0 * d["valu_lane_util"]
            f64_flops = (m.get("SQ_INSTS_VALU_ADD_F64", 0) + m.get("SQ_INSTS_VALU_MUL_F64", 0) + 2.0 * m["SQ_INSTS_VALU_FMA_F64"]) * lanes
            f32_flops = (m.get("SQ_INSTS_VALU_ADD_F32", 0) + m.get("SQ_INSTS_VALU_MUL_F32", 0) + 2.0 * m.get("SQ_INSTS_VALU_FMA_F32", 0)) * lanes
            d["executed_f64_tflops"] = f64_flops / (d["avg_ms"] * 1e-3) / 1e12
            d["executed_f32_tflops"] = f32_flops / (d["avg_ms"] * 1e-3) / 1e12
            d["f64_flops_per_launch"] = f64_flops
            d["f32_flops_per_launch"] = f32_flops
        d["valu_busy_from_mix"] = (4.0 * f64 + 16.0 * t64 + 8.0 * t32 + 2.0 * rest) / (1024.0 * m["GRBM_GUI_ACTIVE"] / 8.0)
    if "SQ_WAIT_ANY" in m and m.get("SQ_WAVE_CYCLES"):
        d["wait_any_share"] = m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"]   # wave cycles parked on a counter
    if "SQ_WAIT_INST_ANY" in m and m.get("SQ_WAVE_CYCLES"):
        d["issue_stall_share"] = m["SQ_WAIT_INST_ANY"] / m["SQ_WAVE_CYCLES"]
    if "SQ_WAVES" in m and "SQ_INSTS_VALU" in m:
        d["valu_per_wave"] = m["SQ_INSTS_VALU"] / m["SQ_WAVES"]
    if "SQ_WAVES" in m and "SQ_INSTS_SALU" in m:
        d["salu_per_wave"] = m["SQ_INSTS_SALU"] / m["SQ_WAVES"]
    out[name] = d
    print(name)
    for k, v in d.items():
        print(f"   {k:24s} {v:.6g}" if isinstance(v, float) else f"   {k:24s} {v}")
if "--json" in sys.argv:
    json.dump(out, open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1)
if "--measured" in sys.argv:
    # profiles/measured_counters.json: what bench.py quotes for the dominant kernel of a workload.
    # HBM bytes per launch = WRITE_SIZE + 2 x FETCH_SIZE (both KiB; on gfx950 FETCH_SIZE tallies the
    # 128-byte requests of wide coalesced loads at 64 bytes).
    path, workload = sys.argv[sys.argv.index("--measured") + 1], sys.argv[sys.argv.index("--measured") + 2]
    name = max((k for k in out if out[k].get("avg_ms")), key=lambda k: out[k]["avg_ms"])
    d = out[name]
    try:
        allw = json.load(open(path))
    except Exception:
        allw = {}
    allw[workload] = {
        "kernel": name,
        "avg_ms_rocprof": d["avg_ms"],
        "hbm_bytes_per_launch": int((d.get("WRITE_SIZE", 0.0) + 2.0 * d.get("FETCH_SIZE", 0.0)) * 1024),
        "write_bytes": int(d.get("WRITE_SIZE", 0.0) * 1024),
        "fetch_bytes_corrected": int(2.0 * d.get("FETCH_SIZE", 0.0) * 1024),
        "valu_issue_util": d.get("valu_issue_util"),
        "valu_busy_quad_cycles": d.get("valu_busy"),
        "valu_busy": d.get("valu_busy_from_mix"),
        "f64_share_of_valu_insts": d.get("f64_share_of_valu_insts"),
        "valu_lane_util": d.get("valu_lane_util"),
        "executed_f64_tflops": d.get("executed_f64_tflops"),
        "executed_f32_tflops": d.get("executed_f32_tflops"),
        "f64_flops_per_launch": d.get("f64_flops_per_launch"),
        "f32_flops_per_launch": d.get("f32_flops_per_launch"),
        "wait_any_share": d.get("wait_any_share"),
        "issue_stall_share": d.get("issue_stall_share"),
        "valu_per_wavefront": d.get("valu_per_wave"),
        "salu_per_wavefront": d.get("salu_per_wave"),
        "avg_waves_per_simd": d.get("avg_waves_per_simd"),
        "clock_GHz": d.get("clock_GHz"),
        "source": src,
        # bench.py quotes these figures only for the tree they were taken from
        "source_sha256": kernel_source_hash(),
    }
    json.dump(allw, open(path, "w"), indent=1)
