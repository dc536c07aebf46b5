"""Timeline of one training step from a rocprofv3 --kernel-trace CSV (measurement aid).
usage: python tools/timeline.py <kernel_trace.csv> [step_index]
Prints, for one steady-state step: span, per-queue busy time, union busy time, idle gaps, and the kernels in start order."""
import csv, sys, collections

rows = list(csv.DictReader(open(sys.argv[1])))
ks = []
for r in rows:
    ks.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "0"), r.get("Stream_Id", "0")))
ks.sort()
# steps are delimited by the input pack (k_pack_input, or the first layer's launch that contains it: k_conv_thin<.., true>)
starts = [i for i, k in enumerate(ks) if "k_pack_input" in k[2] or ("k_conv_thin" in k[2] and "true" in k[2])]
si = int(sys.argv[2]) if len(sys.argv) > 2 else len(starts) // 2
a, b = starts[si], starts[si + 1]
step = ks[a:b]
t0 = step[0][0]
span = max(k[1] for k in step) - t0
print("step %d: %d kernels, span %.1f us" % (si, len(step), span / 1e3))
busy = collections.defaultdict(float)
for k in step:
    busy[k[4]] += (k[1] - k[0]) / 1e3
print("busy per stream (us):", dict(busy), " sum %.1f" % sum(busy.values()))
ev = sorted([(k[0], 1) for k in step] + [(k[1], -1) for k in step])
act, last, union, over = 0, t0, 0.0, 0.0
for t, d in ev:
    if act > 0:
        union += t - last
    if act > 1:
        over += t - last
    act += d
    last = t
print("union busy %.1f us, idle %.1f us, >=2 kernels concurrently %.1f us" % (union / 1e3, (span - union) / 1e3, over / 1e3))
if len(sys.argv) > 3:
    for k in step:
        nm = k[2].split("(")[0][:44]
        print("%9.1f %8.1f  s%-3s %s" % ((k[0] - t0) / 1e3, (k[1] - k[0]) / 1e3, k[4], nm))
