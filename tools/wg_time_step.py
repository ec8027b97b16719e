"""Workgroup-time of one training step from a rocprofv3 kernel trace: for every kernel of the middle step, duration x workgroups
(capped at 256 CUs' worth when the launch is one wave of workgroups) - the share of the chip a launch holds while it runs, which is
what the other stream loses.  python tools/wg_time_step.py <kernel_trace.csv>"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [int(r["Start_Timestamp"]) for r in rows if "dice_partial" in r["Kernel_Name"]]
k = len(marks) // 2
a, b = marks[k], marks[k + 1]
short = lambda n: n.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0][:70]
agg = collections.OrderedDict()
for r in rows:
    s = int(r["Start_Timestamp"])
    if not (a <= s < b): continue
    d = (int(r["End_Timestamp"]) - s) / 1e3
    wgs = 1
    for ax in "XYZ":
        wgs *= max(1, int(r[f"Grid_Size_{ax}"]) // max(1, int(r[f"Workgroup_Size_{ax}"])))
    key = (r["Stream_Id"], short(r["Kernel_Name"]))
    e = agg.setdefault(key, [0, 0.0, 0.0, 0])
    e[0] += 1; e[1] += d; e[2] += d * wgs; e[3] = max(e[3], wgs)
print(f"step {(b - a) / 1e3:.1f} us")
print(f"{'stream':>6} {'calls':>5} {'sum us':>9} {'wg x ms':>9} {'max wgs':>8}  kernel")
for (st, name), e in sorted(agg.items(), key=lambda kv: -kv[1][2]):
    print(f"{st:>6} {e[0]:5d} {e[1]:9.1f} {e[2] / 1e3:9.1f} {e[3]:8d}  {name}")
