"""Average kernel duration and start-to-previous-end gap per kernel name from a rocprofv3 kernel-trace CSV."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:50]) for r in rows)
acc = collections.defaultdict(list)
for (s0, e0, n0), (s1, e1, n1) in zip(ks, ks[1:]):
    acc[n0 + " -> " + n1].append(((e0 - s0) / 1e3, (s1 - e0) / 1e3))
for k, v in sorted(acc.items(), key=lambda kv: -len(kv[1]))[:int(sys.argv[2]) if len(sys.argv) > 2 else 12]:
    print("%-110s n=%4d dur %7.1f gap %6.1f" % (k, len(v), sum(a for a, b in v) / len(v), sum(b for a, b in v) / len(v)))
