"""Per kernel family: SQ wait-state shares, LDS bank-conflict share and the NORMALISED MFMA utilisation, from rocprofv3 --pmc
passes over scripts/profile_frame.py (one eager pass; the tuning launches before it are skipped: only the LAST `n_last`
dispatches count).
usage: pmc_sq_summary.py <out.json> <n_last_dispatches, 0 = the last eager frame> <pass dir or counter_collection.csv> [more ...]
A pass DIRECTORY (rocprofv3 -d) also gives the pass's kernel trace, i.e. each dispatch's duration:
  mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (sum of kernel ns * 2.4 GHz * 4 SIMDs * 256 CUs)
(the counter counts busy cycles of every SIMD's matrix pipe, MI355X_MICROARCH.md 'Per-instruction cycle constants'; 2.4 GHz is
the part's maximum clock, so the fraction reads LOW by whatever the chip clocks down under load)."""
import csv
import glob
import json
import os
import sys

FAMILIES = (("conv_halo_kernel", "conv_halo"), ("conv_c64", "conv_halo"), ("conv_gemm", "conv_gemm"), ("tail_kernel", "fused_tail"), ("attention_", "attention"),
            ("gn_", "groupnorm"), ("splitk", "splitk_reduce"), ("layernorm", "layernorm"))
CLOCK_HZ, SIMDS = 2.4e9, 4 * 256


def family(name):
    for key, fam in FAMILIES:
        if key in name:
            return fam
    return "other"


out = {}
n_last = int(sys.argv[2])
for arg in sys.argv[3:]:
    paths = [arg] if arg.endswith(".csv") else glob.glob(os.path.join(arg, "**", "*_counter_collection.csv"), recursive=True)
    traces = [] if arg.endswith(".csv") else glob.glob(os.path.join(arg, "**", "*_kernel_trace.csv"), recursive=True)
    for path in paths:
        rows = list(csv.DictReader(open(path)))
        ids = sorted({int(r["Dispatch_Id"]) for r in rows})
        if n_last > 0:
            ids = ids[-n_last:]
        else:  # 0: the last eager frame = everything from the last preprocess_rgb dispatch on
            start = max(int(r["Dispatch_Id"]) for r in rows if "preprocess" in r["Kernel_Name"])
            ids = [i for i in ids if i >= start]
        keep = set(ids)
        has_mfma = False
        for r in rows:
            if int(r["Dispatch_Id"]) in keep:
                d = out.setdefault(family(r["Kernel_Name"]), {})
                d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
                has_mfma = has_mfma or r["Counter_Name"] == "SQ_VALU_MFMA_BUSY_CYCLES"
        if has_mfma and traces:  # durations of the SAME pass (a profiled pass runs slower than an un-profiled one)
            for r in csv.DictReader(open(traces[0])):
                if int(r["Dispatch_Id"]) in keep:
                    d = out.setdefault(family(r["Kernel_Name"]), {})
                    d["kernel_ns_in_the_mfma_pass"] = d.get("kernel_ns_in_the_mfma_pass", 0.0) + (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for fam, d in out.items():
    wc = d.get("SQ_WAVE_CYCLES")
    if wc:
        for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS"):
            if k in d:
                d[k + "/WAVE_CYCLES"] = round(d[k] / wc, 4)
    if d.get("SQ_LDS_IDX_ACTIVE"):
        d["LDS_BANK_CONFLICT/IDX_ACTIVE"] = round(d.get("SQ_LDS_BANK_CONFLICT", 0.0) / d["SQ_LDS_IDX_ACTIVE"], 4)
    if d.get("kernel_ns_in_the_mfma_pass") and "SQ_VALU_MFMA_BUSY_CYCLES" in d:
        d["mfma_busy_frac"] = round(d["SQ_VALU_MFMA_BUSY_CYCLES"] / (d["kernel_ns_in_the_mfma_pass"] * 1e-9 * CLOCK_HZ * SIMDS), 4)
json.dump(out, open(sys.argv[1], "w"), indent=1, sort_keys=True)
print(json.dumps({f: {k: v for k, v in d.items() if "/" in k or k == "mfma_busy_frac"} for f, d in out.items()}, sort_keys=True))
