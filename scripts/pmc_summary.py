"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes: fabric-side bytes per launch of the implicit-GEMM kernel
(profiles/pmc_conv_gemm.json, read by bench.py for `roofline.traffic`) and, per kernel family, bytes and GB/s.
usage: pmc_summary.py <fetch pass dir or csv> <write pass dir or csv> <ops.json> <out.json>
Only the dispatches of the LAST eager frame are used (the tuning launches before it are skipped).
Bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950: FETCH_SIZE tallies 128-byte requests at 64 bytes -> doubled,
MI355X_MICROARCH.md 'HBM'; Infinity-Cache hits are counted by these fabric-side counters, so this is an upper bound on DRAM
traffic).  GB/s of a family = its bytes / the sum of its kernels' durations in the same pass (pass directories only)."""
import csv
import glob
import json
import os
import sys

FAMILIES = (("conv_halo_kernel", "conv_halo"), ("conv_c64", "conv_halo"), ("conv_gemm", "conv_gemm"), ("tail_kernel", "fused_tail"), ("attention_", "attention"),
            ("gn_", "groupnorm"), ("splitk", "splitk_reduce"), ("layernorm", "layernorm"))


def family(name):
    for key, fam in FAMILIES:
        if key in name:
            return fam
    return "other"


def files(arg, suffix):
    return [arg] if arg.endswith(".csv") else glob.glob(os.path.join(arg, "**", "*" + suffix), recursive=True)


def per_dispatch(arg, counter):
    per, name = {}, {}
    for path in files(arg, "_counter_collection.csv"):
        for r in csv.DictReader(open(path)):
            if r.get("Counter_Name") == counter:  # several rows per dispatch (one per counter instance) are summed
                i = int(r["Dispatch_Id"])
                per[i] = per.get(i, 0.0) + float(r["Counter_Value"])
                name[i] = r["Kernel_Name"]
    dur = {}
    if not arg.endswith(".csv"):
        for path in files(arg, "_kernel_trace.csv"):
            for r in csv.DictReader(open(path)):
                dur[int(r["Dispatch_Id"])] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    return per, name, dur


ops = json.load(open(sys.argv[3]))
n_conv = sum(1 for m in ops if m["op"] == "conv")
alg = sum(m.get("wbytes", 0) + 2 * m["M"] * (m["K"] // (m["ks"] ** 2)) + 2 * m["M"] * m["N"] for m in ops if m["op"] == "conv")
fetch, fname, fdur = per_dispatch(sys.argv[1], "FETCH_SIZE")
write, wname, wdur = per_dispatch(sys.argv[2], "WRITE_SIZE")
is_conv = lambda n: ("conv_gemm" in n and "group" not in n) or "conv_halo_kernel" in n or "conv_c64" in n  # noqa: E731
fc = [fetch[i] for i in sorted(fetch) if is_conv(fname[i])][-n_conv:]
wc = [write[i] for i in sorted(write) if is_conv(wname[i])][-n_conv:]
fetch_b = 2.0 * sum(fc) * 1024 / len(fc)
write_b = sum(wc) * 1024 / len(wc)
res = {"hbm_bytes_per_launch": fetch_b + write_b, "fetch_bytes_per_launch_corrected": fetch_b,
       "write_bytes_per_launch": write_b, "launches": len(fc), "algorithmic_bytes_per_launch": alg / n_conv,
       "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over scripts/profile_frame.py, last eager "
                 "frame only; bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 per launch; algorithmic = fp16 weights + input + output"}
# per family, over the last eager frame (= the dispatches from the last preprocess kernel on)
fam = {}
for per, name, dur, key in ((fetch, fname, fdur, "fetch"), (write, wname, wdur, "write")):
    ids = sorted(per)
    start = max((i for i in ids if "preprocess" in name[i]), default=ids[0])
    for i in ids:
        if i < start:
            continue
        d = fam.setdefault(family(name[i]), {"fetch_bytes": 0.0, "write_bytes": 0.0, "ns_fetch_pass": 0, "ns_write_pass": 0, "launches": 0})
        d[key + "_bytes"] += per[i] * 1024 * (2.0 if key == "fetch" else 1.0)
        d["ns_" + key + "_pass"] += dur.get(i, 0)
        if key == "fetch":
            d["launches"] += 1
for d in fam.values():
    ns = max(d["ns_fetch_pass"], d["ns_write_pass"])
    d["fabric_GBps"] = round((d["fetch_bytes"] + d["write_bytes"]) / ns, 1) if ns else None  # bytes / ns = GB/s
res["families"] = fam
json.dump(res, open(sys.argv[4], "w"), indent=1)
print(json.dumps({k: v for k, v in res.items() if k != "families"}))
print(json.dumps({f: {"fabric_GBps": d["fabric_GBps"], "MB": round((d["fetch_bytes"] + d["write_bytes"]) / 1e6, 1)} for f, d in fam.items()}))
