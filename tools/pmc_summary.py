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
# Units (MI355X_MICROARCH.md, per-instruction constants): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count QUAD-cycles, SQ_VALU_MFMA_BUSY_CYCLES
# counts cycles - so the VALU_MFMA_BUSY column below (counter / SQ_WAVE_CYCLES) is a RELATIVE INDEX, not a utilisation: a register-only
# v_mfma_f32_16x16x32_bf16 loop (vs_debug_mfma_rate, one wave per SIMD) reads 0.497 on this scale.  The last column rescales it by that
# reading: "fraction of what a pure MFMA loop shows per wave cycle".
print("columns other than SQ_WAVE_CYCLES: counter / SQ_WAVE_CYCLES.  VALU_MFMA_BUSY is a relative index (cycles over quad-cycles; a pure MFMA loop reads 0.497);")
print("MFMA_vs_pure_loop = that index / 0.497.")
print("kernel".ljust(72) + " ".join(c.replace("SQ_", "")[:16].rjust(17) for c in ctrs) + " MFMA_vs_pure_loop".rjust(19))
for n in names:
    wc = agg[n].get("SQ_WAVE_CYCLES", 1) or 1
    print(n.ljust(72) + " ".join((f"{agg[n].get(c, 0) / wc:17.3f}" if c != "SQ_WAVE_CYCLES" else f"{agg[n][c]:17.3e}") for c in ctrs) +
          f"{agg[n].get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / wc / 0.497:19.3f}")
