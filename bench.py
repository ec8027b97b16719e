"""bench.py - headline benchmark of the hot path (BASELINE.json metric) on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run)

A step = one training step of U-Net/ResNet-34 (2 classes) on a batch of 32 synthetic 256x256 slices that
are already resident in HBM: weight-copy refresh, forward (train-mode BN), DiceLoss(normalization="none"),
backward (all weights trainable), gradient all-reduce over RCCL when N > 1, fused AdamW, OneCycleLR step -
i.e. VolSeg2dTrainer._train_one_batch (reference vol_seg_2d_trainer.py:419-432) with bf16 activations and
fp32 master weights.  Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

REPO = Path(__file__).resolve().parent
sys.path.insert(0, str(REPO))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np
import torch

FLOP_PER_SLICE_FWD_BWD_256 = 46.22e9   # 3 * 7.721 GMAC - stem dgrad, x2
MFMA_PEAK_BF16_TFLOPS = 2500.0         # MI355X dense bf16 (MI355X_MICROARCH.md, chip-level parameters)
HBM_PEAK_GBS = 8000.0


def synth_batch(batch: int, size: int, classes: int, seed: int):
    """Slices of a seeded blurred-noise uint8 volume + labels by thresholding the same field."""
    rng = np.random.default_rng(seed)
    v = rng.standard_normal((batch, size, size)).astype(np.float32)
    for ax in (1, 2):
        for _ in range(3):
            v = (np.roll(v, 1, ax) + v + np.roll(v, -1, ax)) / 3.0
    v = (v - v.mean()) / v.std()
    u8 = np.clip(128 + 40 * v, 0, 255).astype(np.uint8)
    qs = np.quantile(v, [0.65] if classes == 2 else list(np.linspace(0, 1, classes + 1)[1:-1]))
    lab = np.digitize(v, qs).astype(np.int64)
    x = ((u8.astype(np.float32) / 255) - 0.449) / 0.226   # data/datasets.py:63-69
    return torch.from_numpy(x).unsqueeze(1), torch.from_numpy(lab)


def load_traffic(kernel_name: str, workload: str = "train"):
    """HBM bytes per launch of the dominant kernel from the newest committed PMC passes (profiles/r*_pmc_traffic.json, made by
    tools/pmc_traffic.py from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of this command with the gfx950 FETCH_SIZE x2
    correction), or None.  Counters cannot be read from inside the timed process, so the figure is a committed measurement:
    the entry carries its source file so that it cannot pass for a live one."""
    for f in sorted((REPO / "profiles").glob(f"r*_pmc_traffic{'' if workload == 'train' else '_' + workload}.json"), reverse=True):
        try:
            d = json.loads(f.read_text())
            t = d.get(kernel_name)
        except Exception:
            continue
        if t:
            return dict(t, source=f"profiles/{f.name} (committed rocprofv3 --pmc passes, not this run)")
        # a name PREFIX (the ring kernels: one variant code covers several instantiations, and `roofline.achieved` sums the
        # time and the FLOPs of ALL their launches): launch-weighted mean over the same set of kernels
        rows = [v for k, v in d.items() if isinstance(v, dict) and k.startswith(kernel_name) and v.get("launches")]
        if rows:
            n = sum(v["launches"] for v in rows)
            out = {key: int(sum(v[key] * v["launches"] for v in rows) / n) for key in ("bytes_per_launch", "read_bytes_per_launch", "write_bytes_per_launch")
                   if all(key in v for v in rows)}
            return dict(out, kernels=len(rows), launches=n, correction=rows[0].get("correction"),
                        source=f"profiles/{f.name} (committed rocprofv3 --pmc passes, not this run; launch-weighted over {len(rows)} instantiations)")
    return None


def load_traffic_class(prefixes):
    """Launch-weighted HBM bytes per launch over every kernel of the newest committed PMC passes whose name starts with one of
    `prefixes` (a kernel class served by several instantiations), or None."""
    for f in sorted((REPO / "profiles").glob("r*_pmc_traffic.json"), reverse=True):
        try:
            d = json.loads(f.read_text())
        except Exception:
            continue
        rows = [v for k, v in d.items() if k.startswith(tuple(prefixes)) and v.get("launches")]
        if rows:
            n = sum(v["launches"] for v in rows)
            return {"bytes_per_launch": int(sum(v["bytes_per_launch"] * v["launches"] for v in rows) / n),
                    "read_bytes_per_launch": int(sum(v["read_bytes_per_launch"] * v["launches"] for v in rows) / n),
                    "write_bytes_per_launch": int(sum(v["write_bytes_per_launch"] * v["launches"] for v in rows) / n),
                    "kernels": len(rows), "launches": n, "correction": rows[0].get("correction"),
                    "source": f"profiles/{f.name} (committed rocprofv3 --pmc passes, not this run)"}
    return None


def load_step_traffic():
    """HBM bytes of ONE training step summed over every kernel of the newest committed PMC passes (profiles/r*_pmc_traffic.json:
    rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of `bench.py --no-predict`), or None.  Steps in that run = launches of the stem
    kernel (one per step); the GEMM / MFMA-loop cross-check kernels of the bench are left out."""
    for f in sorted((REPO / "profiles").glob("r*_pmc_traffic.json"), reverse=True):
        try:
            d = json.loads(f.read_text())
        except Exception:
            continue
        steps = (d.get("stem_fwd_bf16_kernel") or {}).get("launches")
        if not steps:
            continue
        rows = {k: v for k, v in d.items() if isinstance(v, dict) and v.get("launches") and not k.startswith(("Cijk", "debug_mfma", "at::", "void at::"))}
        rd = sum(v["read_bytes_per_launch"] * v["launches"] for v in rows.values()) / steps
        wr = sum(v["write_bytes_per_launch"] * v["launches"] for v in rows.values()) / steps
        top = sorted(rows.items(), key=lambda kv: -kv[1]["bytes_per_launch"] * kv[1]["launches"])[:6]
        return {"bytes_per_step": int(rd + wr), "read_bytes_per_step": int(rd), "write_bytes_per_step": int(wr), "steps_in_the_counter_run": steps,
                "largest": [{"kernel": k[:60], "mb_per_step": round(v["bytes_per_launch"] * v["launches"] / steps / 1e6, 1)} for k, v in top],
                "source": f"profiles/{f.name} (committed rocprofv3 --pmc passes, not this run)"}
    return None


def dice_loss(output, target, eps=1e-6):
    """DiceLoss(normalization='none') of the reference (data/pytorch3dunet_losses.py:15-41,89-135)."""
    inter = (output * target).sum((0, 2, 3))
    denom = (output * output).sum((0, 2, 3)) + (target * target).sum((0, 2, 3))
    return 1.0 - torch.mean(2 * (inter / denom.clamp(min=eps)))


def rank_is_zero() -> bool:
    return int(os.environ.get("RANK", "0")) == 0


def usable_cpus() -> int:
    """CPU share of this container: affinity mask, cgroup quota, and the 16-core share of a 1-GPU box."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = Path("/sys/fs/cgroup/cpu.max").read_text().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, 16))


def log(msg):
    print(f"[bench] {msg}", file=sys.stderr, flush=True)


def cpu_baseline(batch: int, steps: int):
    """The oracle (pure-torch CPU fp32 restatement of the reference path) timed on this host: reported
    baseline, not the target.  Bounded sample: `steps` training steps at the reference's default batch 12."""
    from oracle.unet_resnet34_torch import seeded_oracle
    cores = usable_cpus()
    torch.set_num_threads(cores)
    log(f"cpu baseline on {cores} threads")
    net = seeded_oracle(2, 0, perturb_bn=False)
    net.train()
    opt = torch.optim.AdamW(net.parameters(), lr=1e-3)
    x, lab = synth_batch(batch, 256, 2, seed=99)
    t = torch.nn.functional.one_hot(lab, 2).permute(0, 3, 1, 2).float()
    def step():
        opt.zero_grad()
        loss = dice_loss(net(x), t)
        loss.backward()
        opt.step()
    step()
    log("cpu baseline warm-up step done")
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
        log(f"cpu baseline step done at {time.perf_counter() - t0:.1f}s")
    dt = time.perf_counter() - t0
    return {"value": round(batch * steps / dt, 3), "unit": "slices/s", "cores": cores, "kind": "port",
            "sample": f"{steps} fwd+bwd+AdamW steps, batch {batch} (reference default), 256x256, fp32, torch CPU "
                      f"({torch.get_num_threads()} threads) after 1 warm-up step"}


def cpu_baseline_predict():
    """BASELINE.md section 3, prediction legs: the oracle's _predict_single_axis (batch 4, the reference's default) on 16
    slices of 256^2 (2 classes) and 512^2 (4 classes), and ONE _merge_vols_in_mem-equivalent NumPy merge
    (vol_seg_2d_predictor.py:90-98: argmax over the two slots -> int64 index volume -> two take_along_axis) at 256^3 and
    512^3; extrapolated linearly to the full volumes and marked as extrapolated."""
    from oracle import predictor_numpy as P
    from oracle.unet_resnet34_torch import seeded_oracle
    cores = usable_cpus()
    torch.set_num_threads(cores)
    out = {"cores": cores, "kind": "port", "batch": 4}
    rng = np.random.default_rng(5)
    for size, classes in ((256, 2), (512, 4)):
        net = seeded_oracle(classes, 0).eval()
        vol = rng.integers(0, 256, (16, size, size), dtype=np.uint8)
        P.predict_single_axis(net, vol[:4], 0, batch_size=4)            # warm-up
        t0 = time.perf_counter()
        P.predict_single_axis(net, vol, 0, batch_size=4)
        dt = time.perf_counter() - t0
        out[f"predict_{size}sq_slices_per_s"] = round(16 / dt, 2)
        log(f"cpu predict {size}^2: {16 / dt:.2f} slices/s")
    for cube in (256, 512):
        n = cube ** 3
        prob = rng.random((2, cube, cube, cube), dtype=np.float32).astype(np.float16)
        lab = rng.integers(0, 4, (2, cube, cube, cube), dtype=np.uint8)
        t0 = time.perf_counter()
        P.merge_vols_in_mem(prob, lab)
        dt = time.perf_counter() - t0
        out[f"merge_{cube}cube_s"] = round(dt, 3)
        log(f"cpu merge {cube}^3: {dt:.2f} s")
        del prob, lab
    out["extrapolated"] = {
        "predict_256cube_low_2class_s": round(256 / out["predict_256sq_slices_per_s"], 1),
        "predict_512cube_12way_4class_s": round(6144 / out["predict_512sq_slices_per_s"] + 11 * out["merge_512cube_s"], 1),
        "note": "16-slice samples scaled linearly to 256 / 6144 slices, plus 11 merges at the measured 512^3 merge time"}
    out["sample"] = (f"oracle eval forward + softmax/argmax/max-prob, batch 4, 16 slices each of 256^2 (2 classes) and 512^2 (4 classes); "
                     f"one NumPy pairwise merge at 256^3 and at 512^3; torch CPU fp32, {cores} threads")
    return out


def gemm_crosscheck(dev):
    """A measured library bf16 GEMM next to the 2.5 PF vendor peak the roofline fractions use: 8192^3
    torch.matmul (hipBLASLt / rocBLAS under PyTorch-ROCm) on uniform random operands, HIP-event timed."""
    n = 8192
    g = torch.Generator(device=dev).manual_seed(1)
    a = (torch.rand(n, n, device=dev, generator=g) * 2 - 1).to(torch.bfloat16)
    b = (torch.rand(n, n, device=dev, generator=g) * 2 - 1).to(torch.bfloat16)
    for _ in range(3):
        torch.matmul(a, b)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 10
    e0.record()
    for _ in range(reps):
        torch.matmul(a, b)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    tf = 2.0 * n ** 3 / (ms * 1e-3) / 1e12
    out = {"what": "torch.matmul bf16 8192^3 (library GEMM, random operands)", "ms": round(ms, 3), "tflops": round(tf, 1),
           "frac_of_vendor_peak": round(tf / MFMA_PEAK_BF16_TFLOPS, 4)}
    # what a loop of nothing but v_mfma_f32_16x16x32_bf16 (register operands, every CU, 8 waves per SIMD) sustains on this box,
    # with the clock the chip holds meanwhile (MI355X_MICROARCH.md "DVFS give-back"): the practical ceiling under the vendor peak
    from volume_segmantics_amd import _lib
    import ctypes
    tfl, ghz = ctypes.c_double(), ctypes.c_double()
    _lib.check(_lib.lib.vs_debug_mfma_rate(20000, 8, ctypes.byref(tfl), ctypes.byref(ghz)))
    out["mfma_only_loop"] = {"tflops": round(tfl.value, 1), "in_kernel_clock_ghz": round(ghz.value, 3),
                             "frac_of_vendor_peak": round(tfl.value / MFMA_PEAK_BF16_TFLOPS, 4)}
    return out


def synth_volume(n: int, seed: int) -> np.ndarray:
    """uint8 n^3 volume: box-blurred gaussian noise, mean 128 / sd 40, built in float16-free
    chunks so that 512^3 stays cheap on the host."""
    rng = np.random.default_rng(seed)
    v = rng.standard_normal((n, n, n), dtype=np.float32)
    for ax in range(3):
        v = (np.roll(v, 1, ax) + v + np.roll(v, -1, ax)) * (1.0 / 3.0)
    v -= v.mean()
    v *= 40.0 / v.std()
    v += 128.0
    return np.clip(v, 0, 255).astype(np.uint8)


def predict_bench(dev, world, precision, cube: int, classes: int, n_dirs: int, batch: int):
    """Time VolSeg2dPredictor on a synthetic cube: upload once, n_dirs directions sharded over the ranks, packed-key
    max merge, ONE max all-reduce, unpack, labels + probabilities back on the host (the reference API's contract)."""
    from types import SimpleNamespace
    import torch.distributed as dist
    from volume_segmantics_amd.engine import VolSegUnet
    from volume_segmantics_amd.model.operations.vol_seg_2d_predictor import VolSeg2dPredictor
    model = VolSegUnet(classes, device=dev, precision=precision, seed=1)
    with torch.no_grad():  # random-init networks collapse onto one class: centre the head bias on a few slices
        model.eval()
        probe = torch.randn(4, 1, min(cube, 256), min(cube, 256), generator=torch.Generator().manual_seed(7)).to(dev)   # same on every run / rank
        mean_logit = model(probe).mean(dim=(0, 2, 3))
        dict(model.named_parameters())["segmentation_head.0.bias"].sub_(mean_logit)
    if world > 1:
        dist.broadcast(model._flat, 0)
        dist.broadcast(model._bnstate, 0)
    pred = VolSeg2dPredictor.__new__(VolSeg2dPredictor)
    pred.model, pred.num_labels, pred.label_codes = model, classes, {}
    # dedup_directions False: ALL twelve forward passes run in the timed call - the number the earlier rounds reported.  (Four of the
    # reference's twelve directions repeat earlier ones exactly; the predictor's default leaves them out: timed separately below.)
    pred.settings = SimpleNamespace(cuda_device=dev.index, prediction_batch_size=batch, profile_phases=True, dedup_directions=False)
    pred.result_ranks = "rank0"   # every rank holds the merged keys; only rank 0 ships the volume to the host
    vol = synth_volume(cube, seed=5678 if cube == 512 else 1234)
    fn = {1: pred._predict_single_axis, 3: pred._predict_3_ways_max_probs, 12: pred._predict_12_ways_max_probs}[n_dirs]
    fn(vol)  # warm-up at full size: plans, workspaces, pinned staging buffers, key volume, code objects
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    labels, probs = fn(vol)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = tt.item()
    if labels is None:
        labels = np.zeros(1, np.uint8)
    phases = dict(pred.last_timings)     # of the timed call (the instrumented pass below overwrites them)
    dedup = None
    if n_dirs == 12:    # the predictor's default: the four exactly repeated directions are not run - same volume out, 8 passes
        pred.settings.dedup_directions = True
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        labels8, probs8 = fn(vol)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        dt8 = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([dt8], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt8 = tt.item()
        same = None if labels8 is None else bool(np.array_equal(labels8, labels) and np.array_equal(probs8.view(np.uint16), probs.view(np.uint16)))
        dedup = {"seconds": round(dt8, 4), "forward_passes": pred.last_timings.get("directions_run", 0) * cube,
                 "labels_and_probabilities_identical_to_the_12_pass_result": same,
                 "note": "directions 4, 7, 10 and 11 of the reference's call order hold the same slices at the same voxel addresses as directions "
                         "2, 5, 8 and 1 (np.rot90 in the (0, 1) plane maps the Z stack onto the Y stack); the first-wins merge can never take "
                         "a repeat, so 8 passes give the 12-direction volume bit for bit (VolSeg2dPredictor default; dedup_directions: false runs all 12)"}
        pred.settings.dedup_directions = False
    phases_all = [phases]
    if world > 1:       # every rank's phases: with one direction set per rank the slowest one bounds the job
        phases_all = [None] * world
        dist.all_gather_object(phases_all, phases)
    roof = None
    if cube >= 512:     # (every rank runs the instrumented pass - it contains the exchange; rank 0 reports it)
        # roofline of the prediction's dominant kernel: HIP events around every launch of ONE direction (cube slices) on the
        # launch stream, grouped by kernel instantiation
        from volume_segmantics_amd import _lib
        _lib.check(_lib.lib.vs_profile_enable(1))
        pred._predict_single_axis(vol, output_probs=True)
        torch.cuda.synchronize()
        raw = _lib.profile_read_raw(1 << 18)
        _lib.check(_lib.lib.vs_profile_enable(0))
        byvar, total_ms = {}, 0.0
        for kind, tag, ms, fl, by, var in raw:
            total_ms += ms
            if kind == "conv_fwd" and var:
                v = byvar.setdefault(var, [0.0, 0.0, 0])
                v[0] += ms; v[1] += fl; v[2] += 1
        if byvar:
            dv = max(byvar, key=lambda q: byvar[q][0])
            dms, dfl, dn = byvar[dv]
            tname = "unsigned short" if precision == "bf16" else "float"
            code = dv % 10
            name = (f"conv_igemm_kernel<{tname}, {dv // 1000}, {dv // 100 % 10}, {dv // 10 % 10}, 1, 8, 1>" if code == 8 else
                    f"conv_direct_kernel<{tname}, 16, 0>" if code == 4 else
                    f"ring::conv_stream_kernel<{tname}, {dv // 1000}, 2, 8, 4, 2, 2, " if code == 7 else      # (its three epilogue modes share the prefix)
                    f"ring::conv_ring_kernel<{dv // 1000}, " if code == 6 else
                    f"conv_igemm_kernel<{tname}, {dv // 1000}, {dv // 100 % 10}, {dv // 10 % 10}, {code}, 4, 1>")
            ach = dfl / (dms * 1e-3) / 1e12
            tr = load_traffic(name, "predict")
            roof = {"bound": "mfma", "kernel": name, "achieved": round(ach, 1), "peak": MFMA_PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(ach / MFMA_PEAK_BF16_TFLOPS, 4), "traffic": (tr or {}).get("bytes_per_launch"), "traffic_detail": tr,
                    "avg_launch_ms": round(dms / dn, 5), "launches_per_direction": dn,
                    "algorithmic_gflop_per_launch": round(dfl / dn / 1e9, 3), "share_of_kernel_time": round(dms / total_ms, 3),
                    "all_conv_tflops": round(sum(v[1] for v in byvar.values()) / (sum(v[0] for v in byvar.values()) * 1e-3) / 1e12, 1),
                    "note": f"one direction (512 slices, batch {batch}) with HIP events around every launch; flops = 2 x MACs of the layers this instantiation serves"}
    n_slices = n_dirs * cube
    flop = {(256, 2): 15.44e9, (512, 4): 61.92e9}.get((cube, classes), 0.0) * n_slices
    return {"seconds": round(dt, 4), "slices": n_slices, "slices_per_s": round(n_slices / dt, 1),
            "mfma_frac": round(flop / dt / 1e12 / MFMA_PEAK_BF16_TFLOPS / world, 4), "batch": batch,
            "label_hist": np.bincount(labels.ravel(), minlength=classes).tolist(),
            "phases_rank0": {k: round(v, 4) for k, v in phases.items()},
            "phases_per_rank": [{k: round(v, 4) for k, v in ph.items()} for ph in phases_all],
            "includes": "H2D volume upload, all directions, key merge, all-reduce(max), unpack, D2H labels+probs",
            **({"without_the_repeated_directions": dedup} if dedup else {}),
            **({"roofline": roof} if roof else {})}


def merge_bench(dev, cube: int = 512, reps: int = 11):
    """The reference's pairwise max-probability merge (_merge_vols_in_mem, vol_seg_2d_predictor.py:90-98) as vs_merge_maxprob on
    resident volumes: 9 algorithmic bytes per voxel per merge (HBM-bound row B7 of SURVEY.md section 8a)."""
    from volume_segmantics_amd import _lib
    n = cube ** 3
    g = torch.Generator(device=dev).manual_seed(3)
    l0 = torch.randint(0, 4, (n,), dtype=torch.uint8, device=dev, generator=g)
    l1 = torch.randint(0, 4, (n,), dtype=torch.uint8, device=dev, generator=g)
    p0 = torch.rand(n, device=dev, generator=g).half()
    p1 = torch.rand(n, device=dev, generator=g).half()
    st = _lib.stream_ptr()
    _lib.check(_lib.lib.vs_merge_maxprob(_lib.ptr(l0), _lib.ptr(p0), _lib.ptr(l1), _lib.ptr(p1), n, st))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        _lib.check(_lib.lib.vs_merge_maxprob(_lib.ptr(l0), _lib.ptr(p0), _lib.ptr(l1), _lib.ptr(p1), n, st))
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    gbs = 9.0 * n / (ms * 1e-3) / 1e9
    return {"kernel": "merge_maxprob_kernel", "voxels": n, "ms_per_merge": round(ms, 4), "algorithmic_bytes_per_voxel": 9,
            "achieved_gbs": round(gbs, 1), "peak_gbs": HBM_PEAK_GBS, "frac": round(gbs / HBM_PEAK_GBS, 4),
            "eleven_merges_ms": round(ms * 11, 3)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--step-mode", default="auto", choices=["auto", "graph", "eager"],
                    help="graph: replay the recorded step (hipGraphs); eager: enqueue every step call by call; auto: time both "
                         "during the untimed set-up and keep the faster one on this box")
    ap.add_argument("--cpu-steps", type=int, default=10)
    ap.add_argument("--no-predict", action="store_true", help="skip the 512^3 12-direction / 256^3 predict measurements")
    ap.add_argument("--grad-transport", default="auto", choices=["auto", "fp32", "bf16"],
                    help="dtype of the gradient buckets on the wire at N > 1 (auto: the compute precision - bf16 halves the bytes per "
                         "xGMI link, SURVEY.md section 8e; the sum carries <= 2 * N * 2^-8 * max|g| of rounding, tests/test_dist_gloo.py)")
    ap.add_argument("--encoder", default="resnet34", choices=["resnet18", "resnet34", "resnet50", "resnext50_32x4d", "efficientnet-b3", "efficientnet-b4", "timm-resnest50d", "timm-resnest101e"])
    ap.add_argument("--topology", default="unet", choices=["unet", "unetplusplus", "linknet", "fpn", "deeplabv3plus", "deeplabv3", "manet", "pan"],
                    help="with --encoder / --size / --classes: other rows of the model matrix, e.g. BASELINE configs[3] = "
                         "--topology unetplusplus --encoder resnet50 --size 512 --classes 4 (not the headline metric: no FLOP model)")
    ap.add_argument("--size", type=int, default=256, help="slice height = width")
    ap.add_argument("--classes", type=int, default=2)
    ap.add_argument("--per-unit", default="", help="write a per-layer kernel-time table (instrumented steps) to this file")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    # VOLSEG_BENCH_REHEARSE=gloo: every rank on GPU 0 with the gloo backend - a correctness rehearsal of the N > 1 code path on
    # a one-GPU box (the numbers mean nothing); the real run is one rank per GPU over RCCL
    rehearse = os.environ.get("VOLSEG_BENCH_REHEARSE", "")
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    if world > 1:
        if rehearse:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group(rehearse)
        else:
            dist.init_process_group("nccl", device_id=dev)

    from volume_segmantics_amd import _lib
    from volume_segmantics_amd.engine import VolSegUnet

    headline = (args.encoder, args.topology, args.size, args.classes, args.batch) == ("resnet34", "unet", 256, 2, 32)
    if not headline:
        args.no_predict = True      # the prediction legs are the headline configuration's
    model = VolSegUnet(args.classes, device=dev, precision=args.precision, seed=0, encoder=args.encoder, topology=args.topology)
    if world > 1:
        dist.broadcast(model._flat, 0)
        dist.broadcast(model._bnstate, 0)
        model.dp_group = dist.group.WORLD
        if args.grad_transport == "bf16" or (args.grad_transport == "auto" and args.precision == "bf16"):
            model.dp_grad_dtype = torch.bfloat16
        model.dropout_seed += 1000003 * rank
    x, lab = synth_batch(args.batch, args.size, args.classes, seed=1234 + rank)   # every rank: its own shard of the global batch
    x = x.to(dev)
    # one-hot targets as prepare_training_batch hands them over (utilities/base_data_utils.py:150-158): NCHW uint8, contiguous
    target = torch.nn.functional.one_hot(lab, args.classes).permute(0, 3, 1, 2).to(dev, torch.uint8).contiguous()
    opt = model.fused_adamw(lr=1e-4, fuse_step_into_backward=True)   # single GPU: the step hides under the backward pass
    setup_steps = 3   # untimed, before the warm-up: lazy initialisation + recording the step's two hipGraphs (one per weight set)
    total = setup_steps + args.warmup + args.steps + 13 * 2 * (max(4, args.steps) + 1) + 64
    sched = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=1e-3, total_steps=total + 1, pct_start=0.3)
    model.train()

    from volume_segmantics_amd.data.losses import HipDiceLoss
    criterion = HipDiceLoss()

    def step_eager():   # the reference's _train_one_batch, call by call (vol_seg_2d_trainer.py:419-432)
        opt.zero_grad()
        out = model(x)
        loss = criterion(out, target)
        loss.backward()
        opt.step()
        sched.step()
        return loss

    use_graph = args.step_mode != "eager"

    def step():
        # the same step as replayed hipGraphs (VolSegUnet.fused_train_step; N > 1: with the bucket all-reduces between them)
        if use_graph and model.can_fuse_step(opt, x, target):
            loss = model.fused_train_step(x, target, opt, eps=criterion.epsilon, clone_loss=False)
            sched.step()
            return loss
        return step_eager()

    log(f"model ready on {dev}, setting up")
    for _ in range(setup_steps):
        step()
    torch.cuda.synchronize()
    graph_mode = bool(use_graph and model._steps and all(g is not None for st in model._steps.values() for g in st["graphs"].values()))
    # Untimed set-up, continued: bursts of `steps` back-to-back steps (the shape of the timed region) until two consecutive
    # bursts agree within 2 %.  A process's first bursts run 1.5-2.5x slower than its steady state (measured:
    # tools/stream_probe.py - the first 20-step burst after start-up takes 8.5-12.6 ms per step, the following ones 4.83; half a
    # second of earlier activity brings the first burst to 7.0, 1.5 s to 5.0): a training run leaves that transient behind in
    # its first second, a 20-step timed region would consist of nothing else.  The same bursts rank the two step modes on this
    # box: replayed graphs cost the host ~0.4 ms per step against ~2 ms call by call, but order the two streams range by range
    # instead of unit by unit (~2 % more GPU time).
    def burst(fn, k):
        fn(); torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(k):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / k * 1e3
    autotune = {"bursts_ms": []}
    modes = [("graph", step)] if graph_mode else []
    if not graph_mode or args.step_mode == "auto":
        modes.append(("eager", step_eager))
    hist = {nm: [] for nm, _ in modes}
    for i in range(12):
        for nm, fn in modes:
            hist[nm].append(round(burst(fn, max(4, args.steps)), 4))
        if world > 1:
            if i >= 3:      # a fixed count when ranks run collectives inside the steps: every rank must leave the loop together
                break
        elif i >= 1 and all(abs(h[-1] - h[-2]) <= 0.02 * h[-1] for h in hist.values()):
            break
    autotune["bursts_ms"] = hist
    if graph_mode and args.step_mode == "auto":
        last = [hist["graph"][-1], hist["eager"][-1]]
        if world > 1:       # one decision for the whole job: the slowest rank's burst times
            tt = torch.tensor(last, device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            last = tt.tolist()
        use_graph = graph_mode = last[0] <= last[1]
        autotune["picked"] = "graph" if use_graph else "eager"
    log(f"setup done (replaying recorded step graphs: {graph_mode} {autotune}); warming up")
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    log("warm-up done")
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    t_enq = time.perf_counter() - t0          # the host is done enqueueing; the GPU may still be running
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = tt.item()
    final_loss = float(loss.detach())
    log(f"timed region: {elapsed / args.steps * 1e3:.3f} ms/step (host enqueue {t_enq / args.steps * 1e3:.3f} ms/step)")
    # the same steps call by call (no graph): what the host costs when it enqueues ~430 launches per step itself
    eager = {}
    if world == 1:
        n_e = 10
        for _ in range(2):
            step_eager()
        torch.cuda.synchronize()
        te0 = time.perf_counter()
        for _ in range(n_e):
            step_eager()
        te_enq = time.perf_counter() - te0
        torch.cuda.synchronize()
        eager = {"ms_per_step": round((time.perf_counter() - te0) / n_e * 1e3, 4), "host_enqueue_ms": round(te_enq / n_e * 1e3, 4),
                 "steps": n_e}
        log(f"call-by-call: {eager['ms_per_step']} ms/step, host enqueue {eager['host_enqueue_ms']} ms/step")

    # ---- roofline block: per-kernel-class HIP-event timing of the same step (separate, instrumented steps) ----
    prof_steps = 3
    side = _lib.lib.vs_get_option(b"side_stream")
    _lib.set_option("side_stream", 0)   # one kernel at a time: clean per-kernel durations (the timed region above overlaps)
    step_eager()
    torch.cuda.synchronize()
    _lib.check(_lib.lib.vs_profile_enable(1))
    for _ in range(prof_steps):
        step_eager()
    torch.cuda.synchronize()
    prof = _lib.profile_read()
    raw = _lib.profile_read_raw()
    _lib.check(_lib.lib.vs_profile_enable(0))
    _lib.set_option("side_stream", side)
    if rank == 0 and args.per_unit:
        names = _lib.unit_names(model._plans[(args.size, args.size)]["handle"])
        agg = {}
        for kind, tag, ms, fl, by, _var in raw:
            a = agg.setdefault((kind, tag), [0.0, 0.0, 0.0])
            a[0] += ms / prof_steps; a[1] += fl / prof_steps; a[2] += by / prof_steps
        rows = sorted(agg.items(), key=lambda kv: -kv[1][0])
        with open(args.per_unit, "w") as f:
            f.write("ms_per_step\tkind\tunit\tTFLOP/s\tGB/s\n")
            for (kind, tag), (ms, fl, by) in rows:
                f.write(f"{ms:.4f}\t{kind}\t{names[tag] if 0 <= tag < len(names) else tag}\t"
                        f"{fl / (ms * 1e-3) / 1e12 if ms else 0:.1f}\t{by / (ms * 1e-3) / 1e9 if ms else 0:.0f}\n")
    log("instrumented steps done")

    predict = {}
    if not args.no_predict:
        del opt, sched
        model.release_plans()     # vs_graph_destroy / vs_unet_destroy for the training plans
        torch.cuda.empty_cache()
        log("predict: 256^3 single axis")
        predict["predict_256cube_low_2class"] = predict_bench(dev, world, args.precision, 256, 2, 1, 32)
        log("predict: 512^3 12 directions")
        predict["predict_512cube_12way_4class"] = predict_bench(dev, world, args.precision, 512, 4, 12, 128)   # eval mode: the batch size does not change the result (128: enough pixel tiles per CU for the persistent convolution kernel)
        log("predict done")
        if rank == 0:
            predict["merge_512cube"] = merge_bench(dev)

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        slices_per_s = args.batch * world * args.steps / elapsed
        mfma_kinds = ("conv_fwd", "conv_dgrad", "conv_wgrad")
        # dominant kernel symbol: forward and dgrad share conv_igemm_kernel<T,BN,PT,TAPS,STRIDE>; group its launches by instantiation
        byvar = {}
        for kind, tag, ms, fl, by, var in raw:
            if kind in ("conv_fwd", "conv_dgrad") and var:
                v = byvar.setdefault(var, [0.0, 0.0, 0])
                v[0] += ms; v[1] += fl; v[2] += 1
        dvar = max(byvar, key=lambda k: byvar[k][0])
        tname = "unsigned short" if args.precision == "bf16" else "float"
        code = dvar % 10
        dom_name = (f"conv_igemm_kernel<{tname}, {dvar // 1000}, {dvar // 100 % 10}, {dvar // 10 % 10}, 1, 8, 1>" if code == 8 else
                    f"ring::conv_ring_kernel<{dvar // 1000}, " if code == 6 else
                    f"ring::conv_stream_kernel<{tname}, {dvar // 1000}, 2, 8, 4, 2, 2, " if code == 7 else
                    f"conv_igemm_kernel<{tname}, {dvar // 1000}, {dvar // 100 % 10}, {dvar // 10 % 10}, {code}, 4, 1>")
        dom_ms, dom_fl, dom_calls = byvar[dvar]
        ach = dom_fl / (dom_ms * 1e-3) / 1e12 if dom_ms > 0 else 0.0
        conv_ms = sum(prof[k]["ms"] for k in mfma_kinds) / prof_steps
        conv_fl = sum(prof[k]["flops"] for k in mfma_kinds) / prof_steps
        breakdown = {k: {"ms_per_step": round(v["ms"] / prof_steps, 4), "launches_per_step": v["calls"] // prof_steps,
                         **({"tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2)} if v["flops"] and v["ms"] else {}),
                         **({"gbs": round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1)} if v["bytes"] and v["ms"] else {})}
                     for k, v in prof.items()}
        out = {
            "metric": ("slices/sec fwd+bwd U-Net/ResNet-34 256x256 bf16 batch 32 (training step incl. loss, AdamW)" if headline else
                       f"slices/sec fwd+bwd {args.topology}/{args.encoder} {args.size}x{args.size} {args.precision} batch {args.batch} "
                       f"{args.classes}-class (training step incl. loss, AdamW) - NOT the headline configuration"),
            "value": round(slices_per_s, 2), "unit": "slices/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.precision, "data": "synthetic",
            "config": {"workload": ("configs[1]: synthetic 256^3-style 1-ch slices, 2-class, U-Net/ResNet-34, batch 32 per GPU, "
                                    "train step (fwd + DiceLoss + bwd + AdamW + OneCycleLR)") if headline else
                                   (f"synthetic 1-ch slices, {args.classes}-class, {args.topology}/{args.encoder}, batch {args.batch} per GPU, "
                                    "train step (fwd + DiceLoss + bwd + AdamW + OneCycleLR)"),
                       "global_batch": args.batch * world, "slice": f"{args.size}x{args.size}", "parallelism": f"dp{world}",
                       "master_weights": "fp32", "final_loss": round(final_loss, 5),
                       **({"grad_allreduce": str(model.dp_grad_dtype).replace("torch.", "") + " buckets inside backward"} if world > 1 else {})},
            "whole_step_mfma_frac": round(slices_per_s / world * FLOP_PER_SLICE_FWD_BWD_256 / 1e12 / MFMA_PEAK_BF16_TFLOPS, 4) if headline else None,
            # where a step's wall time goes: a host- or gap-bound run shows ms_per_step well above the kernel sums
            "step_timing": {
                "mode": "hipGraph replay (forward + DiceLoss + backward + AdamW recorded once per weight set as linear graphs "
                        "per unit range and stream)" if graph_mode else "call by call",
                "autotune_at_setup": autotune,
                "setup_steps_untimed": setup_steps + sum(len(h) * (max(4, args.steps) + 1) for h in autotune["bursts_ms"].values()),
                "host_enqueue_ms": round(t_enq / args.steps * 1e3, 4),
                "main_stream_kernel_sum_ms": round(sum(v["ms"] for k, v in prof.items() if k != "conv_wgrad") / prof_steps, 4),
                "side_stream_kernel_sum_ms": round(prof["conv_wgrad"]["ms"] / prof_steps, 4),
                "gpu_kernel_sum_ms": round(sum(v["ms"] for v in prof.values()) / prof_steps, 4),
                "call_by_call": eager,
                "note": "kernel sums: HIP-event durations of 3 serialised instrumented steps (side stream off; AdamW and the "
                        "weight-copy kernels, ~0.2 ms on the side stream, are not in the event classes); in the timed region the "
                        "weight gradients overlap the caller's stream.  The timed loop does not read the loss back (the "
                        "reference's loop does, loss.item() at vol_seg_2d_trainer.py:220 - one sync per step)."},
            "roofline": {"bound": "mfma", "kernel": dom_name, "achieved": round(ach, 2), "peak": MFMA_PEAK_BF16_TFLOPS,
                         "unit": "TFLOP/s", "frac": round(ach / MFMA_PEAK_BF16_TFLOPS, 4), "traffic": (load_traffic(dom_name) or {}).get("bytes_per_launch"), "traffic_detail": load_traffic(dom_name),
                         "launches_per_step": dom_calls // prof_steps,
                         "avg_launch_ms": round(dom_ms / max(1, dom_calls), 5),
                         "algorithmic_gflop_per_launch": round(dom_fl / max(1, dom_calls) / 1e9, 3),
                         "all_conv_tflops": round(conv_fl / (conv_ms * 1e-3) / 1e12, 2) if conv_ms else 0.0,
                         "note": "sum of algorithmic conv FLOPs of this kernel's launches / sum of their HIP-event durations, "
                                 "3 serialised instrumented steps (side stream off); the timed region overlaps wgrad on a side stream"},
            "kernel_classes": breakdown,
            **predict,
        }
        # the class with the largest time share next to the dominant forward / dgrad kernel (for the headline step: the weight
        # gradients), same definition: algorithmic FLOPs of its launches / sum of their HIP-event durations
        big = max(mfma_kinds, key=lambda k: prof[k]["ms"])
        bms, bfl, bcalls = prof[big]["ms"], prof[big]["flops"], prof[big]["calls"]
        big_name = {"conv_wgrad": "conv_wgrad_ring_kernel / conv_wgrad_bf16_kernel (+ slab_reduce4_kernel)",
                    "conv_fwd": "conv_igemm_kernel (forward launches)", "conv_dgrad": "conv_igemm_kernel (data-gradient launches)"}[big]
        big_prefix = {"conv_wgrad": ("ring::conv_wgrad_ring_kernel", "conv_wgrad_bf16_kernel", "conv_wgrad_kernel"),
                      "conv_fwd": ("conv_igemm_kernel", "ring::conv_ring_kernel", "ring::conv_stream_kernel", "conv_direct_kernel"),
                      "conv_dgrad": ("conv_igemm_kernel", "ring::conv_ring_kernel", "ring::conv_stream_kernel", "conv_direct_kernel")}[big]
        out["roofline_largest_class"] = {
            "bound": "mfma", "class": big, "kernel": big_name, "achieved": round(bfl / (bms * 1e-3) / 1e12, 2) if bms else 0.0,
            "peak": MFMA_PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": round(bfl / (bms * 1e-3) / 1e12 / MFMA_PEAK_BF16_TFLOPS, 4) if bms else 0.0,
            "ms_per_step": round(bms / prof_steps, 4), "launches_per_step": bcalls // prof_steps,
            "share_of_kernel_time": round(bms / sum(v["ms"] for v in prof.values()), 3),
            "traffic": (load_traffic_class(big_prefix) or {}).get("bytes_per_launch"), "traffic_detail": load_traffic_class(big_prefix)}
        out["peak_crosscheck"] = gemm_crosscheck(dev)
        if headline and args.precision == "bf16":
            # the whole step against the HBM roof (average over the step: ~half of what streaming kernels sustain on this part)
            st = load_step_traffic()
            if st:
                tbs = st["bytes_per_step"] / (ms_per_step * 1e-3) / 1e12
                out["whole_step_hbm"] = dict(st, achieved_tbs=round(tbs, 3), peak_tbs=HBM_PEAK_GBS / 1e3, frac=round(tbs / (HBM_PEAK_GBS / 1e3), 4),
                                             note="committed counter bytes of one step / this run's step time; 1 GiB streaming kernels on this part sustain "
                                                  "5.5 TB/s (1 read : 1 write copy), 6.0 (2 : 1 add), 6.6 (fill) - tools/hbm_probe.py; vs_merge_maxprob 5.5")
        if headline:
            step_tflops = slices_per_s / world * FLOP_PER_SLICE_FWD_BWD_256 / 1e12
            loop = out["peak_crosscheck"].get("mfma_only_loop", {}).get("tflops")
            out["whole_step"] = {"achieved_tflops": round(step_tflops, 1), "frac_of_vendor_peak": round(step_tflops / MFMA_PEAK_BF16_TFLOPS, 4),
                                 "frac_of_measured_mfma_loop": round(step_tflops / loop, 4) if loop else None,
                                 "measured_mfma_loop_tflops": loop, "gflop_per_slice_fwd_bwd": round(FLOP_PER_SLICE_FWD_BWD_256 / 1e9, 2)}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(12, args.cpu_steps)
            if not args.no_predict:
                out["cpu_baseline_predict"] = cpu_baseline_predict()
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
