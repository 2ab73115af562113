"""Assemble profiles/rNN_conv_pmc.json from the three rocprofv3 passes of scripts/probe_traffic_pmc.sh
(gpurun_out/pmc_traffic/t1..t3: FETCH_SIZE | WRITE_SIZE | GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY)
over ONE p2 3x3 launch shape (probe_one.py 8 256 200 336 256 3 1 1).  Usage: make_conv_pmc_json.py <dir> <out.json>"""
import csv, glob, json, sys
d, out = sys.argv[1], sys.argv[2]
agg, name = {}, None
for f in sorted(glob.glob(d + "/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if "halo" in r["Kernel_Name"]:
            agg[r["Counter_Name"]] = float(r["Counter_Value"])      # last launch of the kernel in the pass
            name = r["Kernel_Name"].split("(")[0]
ms = None
for f in sorted(glob.glob(d + "/t3*kernel_trace.csv")):
    dd = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in csv.DictReader(open(f)) if "halo" in r["Kernel_Name"]]
    ms = dd[-1]
g = agg.get("GRBM_GUI_ACTIVE", 0.0)
cyc = g / 8.0
fetch_kb, write_kb = agg.get("FETCH_SIZE", 0.0), agg.get("WRITE_SIZE", 0.0)
doc = {
    "kernel": name, "launch": "3x3 256->256 on [8,200,336,256] (fpn_output2 / rpn_head.conv on p2)",
    "command": "scripts/probe_traffic_pmc.sh: rocprofv3 --pmc FETCH_SIZE | --pmc WRITE_SIZE | --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES "
               "SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY (three separate passes, each with --kernel-trace only) -- python scripts/probe_one.py 8 256 200 336 256 3 1 1",
    "FETCH_SIZE_KB_raw": fetch_kb, "WRITE_SIZE_KB_raw": write_kb,
    "correction": "MI355X_MICROARCH.md HBM section: on gfx950 FETCH_SIZE counts 128-B requests as 64 B for wide (16 B/lane) coalesced reads -> doubled; WRITE_SIZE taken as is",
    "hbm_bytes_per_launch": int(2 * fetch_kb * 1024 + write_kb * 1024),
    "algorithmic_bytes_per_launch": 1103000000,
    "sq_counters_same_launch": {
        "SQ_VALU_MFMA_BUSY_CYCLES": agg.get("SQ_VALU_MFMA_BUSY_CYCLES"), "GRBM_GUI_ACTIVE_sum_over_8_XCD": g,
        "SQ_WAIT_INST_ANY": agg.get("SQ_WAIT_INST_ANY"), "SQ_LDS_BANK_CONFLICT": agg.get("SQ_LDS_BANK_CONFLICT"), "launch_ms": ms,
        "clock_GHz": round(cyc / ms / 1e6, 3) if ms else None,
        "mfma_busy_fraction": round(agg.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (cyc * 1024), 4) if cyc else None},
}
json.dump(doc, open(out, "w"), indent=1)
print(json.dumps(doc["sq_counters_same_launch"]), doc["hbm_bytes_per_launch"])
