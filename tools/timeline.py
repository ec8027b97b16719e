"""Per-step stream occupancy from a rocprofv3 --kernel-trace CSV: busy time per stream, idle gaps on the main stream."""
import csv, sys, collections
rows = [(r["Kernel_Name"], int(r["Stream_Id"]), int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: r[2])
marks = [r[2] for r in rows if "dice_partial" in r[0]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(marks) // 2
a, b = marks[k], marks[k + 1]
win = [r for r in rows if a <= r[2] < b]
print(f"step window {(b - a) / 1e6:.3f} ms, {len(win)} kernels")
for s in sorted({r[1] for r in win}):
    w = [r for r in win if r[1] == s]
    print(f"  stream {s}: {len(w)} kernels, busy {sum(r[3] - r[2] for r in w) / 1e6:.3f} ms, first start +{(w[0][2] - a) / 1e6:.3f} ms, last end +{(w[-1][3] - a) / 1e6:.3f} ms")
main = max({r[1] for r in win}, key=lambda s: sum(1 for r in win if r[1] == s))
w = [r for r in win if r[1] == main]
gaps = [(w[i + 1][2] - w[i][3], w[i][0], w[i + 1][0]) for i in range(len(w) - 1)]
pos = [g for g in gaps if g[0] > 0]
print(f"  main-stream idle gaps: {len(pos)} totalling {sum(g[0] for g in pos) / 1e6:.3f} ms")
short = lambda n: n.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")[:48]
for g in sorted(pos, key=lambda g: -g[0])[:8]:
    print(f"    {g[0] / 1e3:7.1f} us  {short(g[1])} -> {short(g[2])}")
