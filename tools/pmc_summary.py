"""Aggregate a rocprofv3 --pmc counter_collection CSV by kernel name (sum over dispatches)."""
import csv, sys, collections, re
path = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.Counter()
with open(path) as f:
    for row in csv.DictReader(f):
        name = re.sub(r"\(anonymous namespace\)::", "", row["Kernel_Name"])
        name = re.sub(r"\(.*", "", name)[:70]
        agg[name][row["Counter_Name"]] += float(row["Counter_Value"])
        calls[name] += 1
names = sorted(agg, key=lambda n: -agg[n].get("SQ_WAVE_CYCLES", 0))[:14]
ctrs = sorted({c for n in names for c in agg[n]})
print("kernel".ljust(72) + " ".join(c.replace("SQ_", "")[:16].rjust(17) for c in ctrs))
for n in names:
    wc = agg[n].get("SQ_WAVE_CYCLES", 1) or 1
    print(n.ljust(72) + " ".join((f"{agg[n].get(c, 0) / wc:17.3f}" if c != "SQ_WAVE_CYCLES" else f"{agg[n][c]:17.3e}") for c in ctrs))
