"""Sum rocprofv3 --pmc SQ counters per kernel family over the LAST `n_last` dispatches of a run of scripts/profile_frame.py
(one eager pass; the tuning launches before it are skipped).
usage: pmc_sq_summary.py <out.json> <n_last_dispatches> <counter_collection.csv> [more csv ...]"""
import csv, json, sys

FAMILIES = (("conv_halo_kernel", "conv_halo"), ("conv_gemm_kernel", "conv_gemm"), ("attention_kernel", "attention"),
            ("gn_", "groupnorm"), ("splitk", "splitk_reduce"), ("layernorm", "layernorm"))


def family(name):
    for key, fam in FAMILIES:
        if key in name:
            return fam
    return "other"


out = {}
n_last = int(sys.argv[2])
for path in sys.argv[3:]:
    rows = list(csv.DictReader(open(path)))
    ids = sorted({int(r["Dispatch_Id"]) for r in rows})[-n_last:]
    keep = set(ids)
    for r in rows:
        if int(r["Dispatch_Id"]) in keep:
            d = out.setdefault(family(r["Kernel_Name"]), {})
            d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
for fam, d in out.items():
    wc = d.get("SQ_WAVE_CYCLES")
    if wc:
        for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS"):
            if k in d:
                d[k + "/WAVE_CYCLES"] = round(d[k] / wc, 4)
    if d.get("SQ_LDS_IDX_ACTIVE"):
        d["LDS_BANK_CONFLICT/IDX_ACTIVE"] = round(d.get("SQ_LDS_BANK_CONFLICT", 0.0) / d["SQ_LDS_IDX_ACTIVE"], 4)
    if d.get("SQ_BUSY_CYCLES") and "SQ_VALU_MFMA_BUSY_CYCLES" in d:
        d["MFMA_BUSY/SQ_BUSY"] = round(d["SQ_VALU_MFMA_BUSY_CYCLES"] / d["SQ_BUSY_CYCLES"], 4)
json.dump(out, open(sys.argv[1], "w"), indent=1, sort_keys=True)
print(json.dumps(out, sort_keys=True))
