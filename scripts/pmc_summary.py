"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into profiles/pmc_conv_gemm.json.
usage: pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> <ops.json> <out.json>
Only the dispatches of the LAST eager frame are used (the tuning launches before it are skipped).
HBM-side bytes per launch of the implicit-GEMM kernel = (2 * FETCH_SIZE + WRITE_SIZE) * 1024
(gfx950: FETCH_SIZE tallies 128-byte requests at 64 bytes -> doubled, MI355X_MICROARCH.md 'HBM';
Infinity-Cache hits are counted by these fabric-side counters, so this is an upper bound on DRAM traffic)."""
import csv, json, sys


def last_frame(path, counter, n_conv):
    rows = [r for r in csv.DictReader(open(path)) if r.get("Counter_Name") == counter and ("conv_gemm_kernel" in r["Kernel_Name"] or "conv_halo_kernel" in r["Kernel_Name"])]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    per = {}
    for r in rows:  # several rows per dispatch (one per counter instance) are summed
        per[int(r["Dispatch_Id"])] = per.get(int(r["Dispatch_Id"]), 0.0) + float(r["Counter_Value"])
    ids = sorted(per)[-n_conv:]
    return [per[i] for i in ids]


ops = json.load(open(sys.argv[3]))
n_conv = sum(1 for m in ops if m["op"] == "conv")
alg = sum(m.get("wbytes", 0) + 2 * m["M"] * (m["K"] // (m["ks"] ** 2)) + 2 * m["M"] * m["N"] for m in ops if m["op"] == "conv")
f = last_frame(sys.argv[1], "FETCH_SIZE", n_conv)
w = last_frame(sys.argv[2], "WRITE_SIZE", n_conv)
fetch_b = 2.0 * sum(f) * 1024 / len(f)
write_b = sum(w) * 1024 / len(w)
res = {"hbm_bytes_per_launch": fetch_b + write_b, "fetch_bytes_per_launch_corrected": fetch_b,
       "write_bytes_per_launch": write_b, "launches": len(f), "algorithmic_bytes_per_launch": alg / n_conv,
       "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over scripts/profile_frame.py, last eager "
                 "frame only; bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 per launch; algorithmic = fp16 weights + input + output"}
json.dump(res, open(sys.argv[4], "w"), indent=1)
print(json.dumps(res))
