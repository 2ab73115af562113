"""Summarise gpurun_out/pmc_halo/*counter_collection.csv (last launch of each conv kernel)."""
import csv, glob, sys
d = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc_halo"
for h in (1, 0):
    agg = {}
    for f in sorted(glob.glob("%s/h%d_*counter_collection.csv" % (d, h))):
        for r in csv.DictReader(open(f)):
            if "halo" in r["Kernel_Name"] or "bf16x3" in r["Kernel_Name"]:
                agg[r["Counter_Name"]] = float(r["Counter_Value"])
    ms = None
    for f in sorted(glob.glob("%s/h%d_GRBM*kernel_trace.csv" % (d, h))):
        dd = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in csv.DictReader(open(f))
              if "halo" in r["Kernel_Name"] or "bf16x3" in r["Kernel_Name"]]
        ms = dd[-1]
    g = agg.get("GRBM_GUI_ACTIVE", 0) / 8
    print("halo" if h else "generic", "ms %.3f clock %.2f GHz mfma_busy %.1f%%" % (ms, g / ms / 1e6, 100 * agg.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (g * 1024)),
          {k: "%.3g" % v for k, v in agg.items()})
