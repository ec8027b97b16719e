"""Per-layer table of ALGORITHMIC vs COUNTER HBM bytes for the convolution kernels of one training step.

  run   (on the GPU box, once under each of `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE`; the LAST step is the one
         whose launch order is written to gpurun_out/layer_seq.json):
            rocprofv3 --pmc FETCH_SIZE --output-format csv -d <dir> -o f -- python3 tools/per_layer_traffic.py run
  merge (anywhere):  python tools/per_layer_traffic.py merge <fetch.csv> <write.csv> gpurun_out/layer_seq.json <out.md>

Algorithmic bytes of a launch = every operand once: input (the forward conv's virtual input - an upsampled source counts at its
stored, quarter size; for a data gradient the incoming gradient), weights, output (for the weight gradient: x, dy and the fp32
dw), in the plan's dtype.  Counter bytes = FETCH_SIZE x 2 (gfx950) + WRITE_SIZE, KiB units (MI355X_MICROARCH.md, HBM).  The
launch order of the step is recorded with the library's event profiler (vs_profile_read_raw: kind, unit, instantiation) and
matched to the counter rows by order within each kernel family."""
import csv, json, re, sys
from pathlib import Path

REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))


def run():
    import torch
    import bench
    from volume_segmantics_amd import _lib
    from volume_segmantics_amd.engine import VolSegUnet
    from volume_segmantics_amd.data.losses import HipDiceLoss
    dev = torch.device("cuda", 0)
    x, lab = bench.synth_batch(32, 256, 2, seed=1234)
    x = x.to(dev)
    t = torch.nn.functional.one_hot(lab, 2).permute(0, 3, 1, 2).to(dev, torch.uint8).contiguous()
    m = VolSegUnet(2, device=dev, precision="bf16", seed=0)
    o = m.fused_adamw(lr=1e-4, fuse_step_into_backward=True)
    crit = HipDiceLoss()
    m.train()
    def step():
        o.zero_grad(); loss = crit(m(x), t); loss.backward(); o.step()
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    _lib.check(_lib.lib.vs_profile_enable(1))
    step()
    torch.cuda.synchronize()
    raw = _lib.profile_read_raw()
    _lib.check(_lib.lib.vs_profile_enable(0))
    plan = m._plans[(256, 256)]
    names = _lib.unit_names(plan["handle"])
    shapes = {t_[0]: t_[1] for t_ in m._table}
    seq = []
    for kind, tag, ms, fl, by, var in raw:
        if kind not in ("conv_fwd", "conv_dgrad", "conv_wgrad", "head") or (kind == "head" and fl == 0):
            continue     # (the backward's "head" record is the dlogits layout sweep: no convolution kernel)
        nm = names[tag].split(" [")[0] if 0 <= tag < len(names) else str(tag)
        c, h, w = (int(v) for v in names[tag].split("[")[1].rstrip("]").split("x")) if 0 <= tag < len(names) else (0, 0, 0)
        wshape = shapes.get(nm)
        seq.append(dict(kind=kind, unit=nm, variant=var, ms=ms, flops=fl, out_chw=[c, h, w], wshape=list(wshape) if wshape else None))
    Path(REPO / "gpurun_out").mkdir(exist_ok=True)
    json.dump(dict(batch=32, esz=2, seq=seq), open(REPO / "gpurun_out" / "layer_seq.json", "w"))
    print(f"recorded {len(seq)} convolution launches of one step")


def alg_bytes(e, batch, esz):
    """every operand once, in bytes"""
    if e["wshape"] is None:
        return None
    cout, cin, k, _ = e["wshape"]
    c, h, w = e["out_chw"]
    if e["unit"].startswith("segmentation_head"):
        c, h, w = cout, 256, 256
    out_px = batch * h * w
    nm = e["unit"]
    stride = 2 if (".0.conv1.weight" in nm or ".0.conv2.weight" in nm and False or "downsample" in nm) and nm.startswith("encoder.layer") and not nm.startswith("encoder.layer1") else 1
    in_px = out_px * stride * stride
    wbytes = cout * cin * k * k * esz
    x_bytes = in_px * cin * esz
    if nm.startswith("decoder.blocks.") and ".conv1.0." in nm:      # cat(upsample(src0), skip): src0 is stored at quarter size
        skip = {0: 256, 1: 128, 2: 64, 3: 64, 4: 0}[int(nm.split(".")[2])]
        x_bytes = out_px * ((cin - skip) / 4 + skip) * esz
    y_bytes = out_px * cout * esz if not nm.startswith("segmentation_head") else out_px * (cout * 4)      # fp32 NCHW logits
    if e["kind"] in ("conv_fwd", "head"):
        return x_bytes + wbytes + y_bytes
    if e["kind"] == "conv_dgrad":       # reads dy (the output-shaped gradient), writes dx (input-shaped; quarter size through an upsample)
        dy = out_px * (16 if nm.startswith("segmentation_head") else cout) * esz
        return dy + wbytes + x_bytes
    dy = out_px * (16 if nm.startswith("segmentation_head") else cout) * esz
    return x_bytes + dy + cout * cin * k * k * 4


FAMILY = {"conv_fwd": ("conv_igemm_kernel", "conv_direct_kernel"), "conv_dgrad": ("conv_igemm_kernel", "conv_direct_kernel"),
          "head": ("conv_head_kernel", "conv_direct_kernel"), "conv_wgrad": ("conv_wgrad_bf16_kernel", "conv_wgrad_kernel")}


def merge(fetch_csv, write_csv, seq_json, out_md):
    meta = json.load(open(seq_json))
    seq = meta["seq"]
    def load(path, counter):
        rows = []
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] == counter:
                name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
                rows.append((int(r["Dispatch_Id"]), re.sub(r"^void ", "", name).split("(")[0], float(r["Counter_Value"])))
        rows.sort()
        return rows
    fam_main = ("conv_igemm_kernel", "conv_direct_kernel", "conv_head_kernel", "ring::conv_ring_kernel", "ring::conv_stream_kernel")
    fam_w = ("conv_wgrad_bf16_kernel", "conv_wgrad_kernel", "ring::conv_wgrad_ring_kernel")
    def last_step(rows, fam, n):
        sel = [r for r in rows if r[1].startswith(fam)]
        return sel[-n:]
    main_seq = [e for e in seq if e["kind"] != "conv_wgrad"]
    w_seq = [e for e in seq if e["kind"] == "conv_wgrad"]
    fr, wr = load(fetch_csv, "FETCH_SIZE"), load(write_csv, "WRITE_SIZE")
    out = []
    for sub, fam in ((main_seq, fam_main), (w_seq, fam_w)):
        f, w = last_step(fr, fam, len(sub)), last_step(wr, fam, len(sub))
        assert len(f) == len(sub) == len(w), (len(f), len(w), len(sub))
        for e, (_, kf, vf), (_, kw, vw) in zip(sub, f, w):
            assert kf == kw, (kf, kw)
            a = alg_bytes(e, meta["batch"], meta["esz"])
            cnt = 2 * vf * 1024 + vw * 1024
            out.append((e, kf, a, cnt))
    with open(out_md, "w") as fo:
        fo.write("| unit | pass | kernel | algorithmic MB | counter MB (2 x FETCH + WRITE) | counter / algorithmic | us | GB/s (counter) |\n|---|---|---|---|---|---|---|---|\n")
        for e, k, a, cnt in out:
            fo.write(f"| {e['unit']} | {e['kind'].replace('conv_', '')} | `{k[:44]}` | {a / 1e6 if a else float('nan'):.1f} | {cnt / 1e6:.1f} | "
                     f"{cnt / a if a else float('nan'):.2f} | {e['ms'] * 1e3:.1f} | {cnt / (e['ms'] * 1e-3) / 1e9 if e['ms'] else 0:.0f} |\n")
        tot_a = sum(a for _, _, a, _ in out if a)
        tot_c = sum(c for _, _, a, c in out if a)
        fo.write(f"\nAll {len(out)} convolution launches of the step: algorithmic {tot_a / 1e9:.2f} GB, counters {tot_c / 1e9:.2f} GB ({tot_c / tot_a:.2f}x).\n")
    print(open(out_md).read()[-300:])


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run()
    else:
        merge(*sys.argv[2:6])
