"""Where a LONE frame's wall time goes on the GPU (rocprofv3 --kernel-trace of the captured one-frame program):

    rocprofv3 --kernel-trace --output-format csv -d /tmp/ft -- python3 scripts/lone_frame.py --no-prefetch --tag trace
    python3 scripts/frame_timeline.py /tmp/ft/*/*_kernel_trace.csv

Takes the LAST complete frame of the trace (preprocess_rgb ... postprocess_rgb), splits its kernels by hardware queue and prints:
span, time with 0 / 1 / 2 queues busy, per queue the kernel count / busy time / gaps between consecutive kernels (histogram), the
gaps around the fork / join edges, and the busy time per kernel family on the critical queue."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
pre = [i for i, r in enumerate(rows) if "preprocess_rgb_kernel" in r["Kernel_Name"]]
post = [i for i, r in enumerate(rows) if "postprocess_kernel" in r["Kernel_Name"]]
assert pre and post
# the last frame whose preprocess AND postprocess are both in the trace
end = post[-1]
start = max(i for i in pre if i < end)
fr = rows[start:end + 1]
t0, t1 = fr[0]["s"], fr[-1]["e"]
span = (t1 - t0) / 1e3
qkey = "Queue_Id" if "Queue_Id" in fr[0] else "Stream_Id"
by_q = collections.defaultdict(list)
for r in fr:
    by_q[r[qkey]].append(r)
print(f"frame: {len(fr)} kernels, span {span / 1e3:.3f} ms, queues {dict((q, len(v)) for q, v in by_q.items())}")
# time with k queues busy
ev = []
for r in fr:
    ev.append((r["s"], 1))
    ev.append((r["e"], -1))
ev.sort()
busy = collections.Counter()
lvl, last = 0, t0
for t, d in ev:
    busy[min(lvl, 2)] += t - last
    last = t
    lvl += d
print("time with 0 / 1 / >=2 kernels running: " + " / ".join(f"{busy[k] / 1e6:.3f} ms" for k in (0, 1, 2)))
fam_of = lambda n: ("conv" if "conv_" in n else "reduce" if "splitk" in n else "gn" if "gn_" in n else "attn" if "attention" in n else  # noqa: E731
                    "tail" if "tail_kernel" in n else "other")
for q, ks in sorted(by_q.items(), key=lambda kv: -len(kv[1])):
    ks.sort(key=lambda r: r["s"])
    b = sum(r["e"] - r["s"] for r in ks)
    gaps = [ks[i + 1]["s"] - ks[i]["e"] for i in range(len(ks) - 1)]
    hist = collections.Counter()
    for g in gaps:
        hist["<1us" if g < 1000 else "1-2us" if g < 2000 else "2-4us" if g < 4000 else "4-10us" if g < 10000 else "10-50us" if g < 50000 else ">50us"] += 1
    big = sorted(gaps, reverse=True)[:12]
    fam = collections.Counter()
    cnt = collections.Counter()
    for r in ks:
        fam[fam_of(r["Kernel_Name"])] += r["e"] - r["s"]
        cnt[fam_of(r["Kernel_Name"])] += 1
    print(f"queue {q}: {len(ks)} kernels, busy {b / 1e6:.3f} ms, first {((ks[0]['s'] - t0) / 1e3):.1f} us, last end {((ks[-1]['e'] - t0) / 1e3):.1f} us, "
          f"sum of gaps {sum(gaps) / 1e6:.3f} ms (median {sorted(gaps)[len(gaps) // 2] / 1e3:.2f} us)")
    print("   gap histogram: " + ", ".join(f"{k}: {hist[k]}" for k in ("<1us", "1-2us", "2-4us", "4-10us", "10-50us", ">50us")))
    print("   largest gaps (us): " + ", ".join(f"{g / 1e3:.1f}" for g in big))
    print("   busy by family (ms | launches | avg us): " + ", ".join(f"{k} {v / 1e6:.3f} | {cnt[k]} | {v / cnt[k] / 1e3:.1f}" for k, v in fam.most_common()))
