"""Attribute rocprofv3 kernel-trace durations of the LAST eager frame to the engine's ops.
usage: analyze_trace.py <kernel_trace.csv> <ops.json>"""
import collections, csv, json, sys

trace, opsf = sys.argv[1], sys.argv[2]
ops = json.load(open(opsf))
ours = ("conv_gemm_kernel", "conv_gemm8_kernel", "conv_gemm_group_kernel", "conv_halo_kernel", "conv_c64", "tail_kernel", "adain_kernel", "splitk_reduce", "gn_stats", "gn_apply", "gn_fused", "layernorm_kernel", "attention_kernel", "attention_lazy", "attention_pair", "attention_lazy", "attention_pair",
        "preprocess_rgb", "sobel_max", "sobel_apply", "add_noise", "lcm_step", "postprocess")
rows = [r for r in csv.DictReader(open(trace)) if any(o in r["Kernel_Name"] for o in ours)]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last eager frame = everything from the last preprocess_rgb kernel on
start = max(i for i, r in enumerate(rows) if "preprocess" in r["Kernel_Name"])
last = rows[start:]
pos = 0
agg = collections.OrderedDict()
busy = 0
total = 0
for m in ops:
    n = {"groupnorm": 2, "sobel_control": 2}.get(m["op"], 1)
    if m["op"] in ("conv", "conv_group") and pos + 1 < len(last) and "splitk_reduce" in last[pos + 1]["Kernel_Name"]:
        n = 2
    if m["op"] == "groupnorm" and "gn_stats" not in last[pos]["Kernel_Name"]:
        n = 1  # one-launch form (small images), or statistics fused into the producer
    ks = last[pos:pos + n]
    pos += n
    total += n
    dur = [int(k["End_Timestamp"]) - int(k["Start_Timestamp"]) for k in ks]
    busy += sum(dur)
    if m["op"] == "conv":
        key = ("conv", m["M"], m["N"], m["K"], m["ks"], m["tile"], m["split"])
        assert any(f in ks[0]["Kernel_Name"] for f in ("conv_gemm", "conv_halo", "conv_c64")), (m, ks[0]["Kernel_Name"][:60])
    elif m["op"] == "conv_group":
        key = ("conv_group", m["members"], m["M"])
        assert "conv_gemm_group" in ks[0]["Kernel_Name"], (m, ks[0]["Kernel_Name"][:60])
    elif m["op"] in ("tail_a", "tail_b"):
        key = (m["op"], m["M"])
        assert "tail_kernel" in ks[0]["Kernel_Name"], (m, ks[0]["Kernel_Name"][:60])
    elif m["op"] == "groupnorm":
        key = ("gn", m["hw"], m["C"])
    elif m["op"] == "layernorm":
        key = ("ln", m["rows"], m["C"])
    elif m["op"] == "attention":
        key = ("attn", m["sq"], m["sk"], m["d"])
        assert "attention" in ks[0]["Kernel_Name"], (m, ks[0]["Kernel_Name"][:60])
    else:
        key = (m["op"],)
    a = agg.setdefault(key, dict(count=0, ns=0, ns2=0, flops=0.0, wbytes=0))
    a["count"] += 1
    a["ns"] += dur[0]
    a["ns2"] += sum(dur[1:])
    a["flops"] += m.get("flops", 0.0)
    a["wbytes"] += m.get("wbytes", 0)
frame_ns = int(last[pos - 1]["End_Timestamp"]) - int(last[0]["Start_Timestamp"])
print(f"frame span {frame_ns/1e6:.2f} ms, kernel busy {busy/1e6:.2f} ms, {total} kernels")
fam = collections.Counter()
for k, a in agg.items():
    fam["conv" if k[0] in ("tail_a", "tail_b", "conv_group") else k[0]] += a["ns"] + a["ns2"]
print({k: round(v / 1e6, 2) for k, v in fam.items()})
print(f"{'op':48s} {'cnt':>4s} {'tot ms':>8s} {'avg us':>8s} {'2nd us':>7s} {'TF/s':>7s} {'wGB/s':>7s}")
for k, a in sorted(agg.items(), key=lambda kv: -(kv[1]["ns"] + kv[1]["ns2"])):
    t = a["ns"]
    print(f"{str(k):48s} {a['count']:4d} {(t + a['ns2'])/1e6:8.3f} {t/a['count']/1e3:8.1f} {a['ns2']/a['count']/1e3:7.1f} "
          f"{a['flops']/max(t,1)/1e3:7.1f} {a['wbytes']/max(t,1):7.1f}")
