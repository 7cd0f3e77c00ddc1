"""Copy a rocprofv3 *_kernel_stats.csv with kernel names truncated to 120 characters (torch's template names run to KBs)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
with open(sys.argv[2], "w") as out:
    out.write("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs\n")
    for r in rows:
        n = r["Name"]
        if len(n) > 120:
            n = n[:117] + "..."
        out.write('"%s",%s,%s,%s,%s,%s,%s\n' % (n.replace('"', "'"), r["Calls"], r["TotalDurationNs"], r["AverageNs"],
                                               r["Percentage"], r["MinNs"], r["MaxNs"]))
