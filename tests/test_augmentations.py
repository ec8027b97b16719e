"""The reference's training augmentations (volume_segmantics/data/augmentations.py:68-101), restated in NumPy
(data/augmentations.py) and as HIP kernels (csrc/augment.hip).  CPU: properties the transforms must have.  GPU (-m gpu): the device
pipeline against the NumPy one with the SAME drawn parameters and displacement fields, transform by transform."""
import numpy as np
import pytest
import torch

from volume_segmantics_amd.data import augmentations as A


def _pair(size=64, seed=0):
    rng = np.random.default_rng(seed)
    f = rng.standard_normal((size, size)).astype(np.float32)
    for ax in (0, 1):
        f = (np.roll(f, 1, ax) + f + np.roll(f, -1, ax)) / 3
    return np.clip(128 + 220 * f, 0, 255).astype(np.uint8), (f > 0.05).astype(np.uint8) + (f > 0.3).astype(np.uint8)


IDENTITY = dict(crop=None, flip_v=False, rot_k=0, transpose=False, distort=None, clahe_clip=0.0, intensity=None)


def _cases(size):
    jit = np.random.default_rng(3).uniform(-4.8, 4.8, (3, 2)).astype(np.float32)
    steps = 1 + np.random.default_rng(4).uniform(-0.3, 0.3, (2, 6))
    return {
        "identity": dict(IDENTITY),
        "crop": dict(IDENTITY, crop=(size // 2 + 5, size // 2 + 5, 0.37, 0.81)),
        "crop_full": dict(IDENTITY, crop=(size, size, 0.5, 0.5)),
        "flip": dict(IDENTITY, flip_v=True),
        "rot1": dict(IDENTITY, rot_k=1), "rot3_t": dict(IDENTITY, rot_k=3, transpose=True), "flip_rot2_t": dict(IDENTITY, flip_v=True, rot_k=2, transpose=True),
        "elastic": dict(IDENTITY, distort=("elastic", jit, 12345)),
        "grid": dict(IDENTITY, distort=("grid", steps[0], steps[1])),
        "optical+": dict(IDENTITY, distort=("optical", 0.7, 0, 0)), "optical-": dict(IDENTITY, distort=("optical", -0.9, 0, 0)),
        "clahe": dict(IDENTITY, clahe_clip=2.7),
        "bc": dict(IDENTITY, intensity=("bc", 1.15, -0.12)), "gamma": dict(IDENTITY, intensity=("gamma", 0.83)),
        "everything": dict(crop=(size - 7, size - 7, 0.2, 0.6), flip_v=True, rot_k=1, transpose=True, distort=("grid", steps[1], steps[0]),
                           clahe_clip=3.3, intensity=("gamma", 1.17)),
    }


def test_numpy_pipeline_properties():
    img, mask = _pair(64)
    c = {k: dict(v, size=64) for k, v in _cases(64).items()}
    a, m = A.apply_params(img, mask, c["identity"])
    assert np.array_equal(a, img) and np.array_equal(m, mask)
    a, m = A.apply_params(img, mask, c["crop_full"])
    assert np.array_equal(a, img) and np.array_equal(m, mask)                      # a full-size window resizes to itself
    a, m = A.apply_params(img, mask, c["flip_rot2_t"])
    assert np.array_equal(a, np.rot90(img[::-1], 2).T) and np.array_equal(m, np.rot90(mask[::-1], 2).T)
    for name, p in c.items():
        a, m = A.apply_params(img, mask, p)
        assert a.shape == img.shape and a.dtype == np.uint8 and m.dtype == np.uint8, name
        assert set(np.unique(m)) <= {0, 1, 2}, name                                 # nearest sampling never invents a label
        if p["distort"] is None and p["crop"] is None:
            assert np.array_equal(np.bincount(m.ravel(), minlength=3), np.bincount(mask.ravel(), minlength=3)), name
    # reflect-101 borders and the known grid-distortion table of all-ones steps (np.linspace includes its end point)
    row = np.arange(4, dtype=np.uint8)[None].repeat(2, 0)
    assert A.remap(row, np.array([[-2, -1, 0, 1, 2, 3, 4, 5, 6]], np.float32), np.zeros((1, 9), np.float32)).tolist() == [[2, 1, 0, 1, 2, 3, 2, 1, 0]]
    mx, _ = A.grid_distortion_maps(96, 96, np.ones(6), np.ones(6))
    assert np.allclose(mx[:3], [0, 19 / 18, 38 / 18]) and mx[-1] == 95.0
    # intensity maps: brightness by max, gamma
    assert A.brightness_contrast_lut(1.0, 0.0).tolist() == list(range(256)) and A.gamma_lut(1.0)[[0, 128, 255]].tolist() == [0, 128, 255]
    assert A.brightness_contrast_lut(1.2, 0.1)[[0, 100, 255]].tolist() == [25, 145, 255]
    eq = A.clahe(img, 40.0)                                                          # a clip limit above every bin: plain tile equalisation
    assert eq.std() > img.std()
    # the draws: every branch is reachable, parameters inside the reference's ranges
    rng = np.random.default_rng(1)
    ps = [A.sample_params(rng, 64) for _ in range(400)]
    kinds = {None if p["distort"] is None else p["distort"][0] for p in ps}
    assert kinds == {None, "elastic", "grid", "optical"} and {None if p["intensity"] is None else p["intensity"][0] for p in ps} == {None, "bc", "gamma"}
    assert 0.4 < np.mean([p["crop"] is not None for p in ps]) < 0.6 and all(32 <= p["crop"][0] <= 64 for p in ps if p["crop"])
    assert all(1 <= p["clahe_clip"] <= 4 for p in ps if p["clahe_clip"]) and all(0.8 <= p["intensity"][1] <= 1.2 for p in ps if p["intensity"])


def test_host_augmenting_dataset_uses_the_pipeline(tmp_path):
    from PIL import Image
    from volume_segmantics_amd.data.datasets import VolSeg2dDataset
    img, mask = _pair(64)
    (tmp_path / "i").mkdir(); (tmp_path / "m").mkdir()
    for k in range(3):
        Image.fromarray(img).save(tmp_path / "i" / f"d_{k}.png"); Image.fromarray(mask).save(tmp_path / "m" / f"s_{k}.png")
    host = VolSeg2dDataset(tmp_path / "i", tmp_path / "m", 64, augment="host", seed=5)
    x, m = host[0]
    assert x.dtype == torch.float32 and x.shape == (1, 64, 64) and m.dtype == torch.uint8
    raw = VolSeg2dDataset(tmp_path / "i", tmp_path / "m", 64, augment="device")
    x, m = raw[1]
    assert x.dtype == torch.uint8 and np.array_equal(x[0].numpy(), img) and np.array_equal(m.numpy(), mask)


@pytest.mark.gpu
@pytest.mark.parametrize("size", [64, 96])
def test_device_pipeline_equals_numpy_pipeline(size):
    """Each transform alone and all together: the kernels reproduce the NumPy pipeline for the same parameters (elastic: the
    same displacement fields, read back from the device) - masks identical, images identical up to a last-bit rounding of the
    bilinear blend on a few pixels."""
    from volume_segmantics_amd.data.gpu_augment import augment_batch
    cases = {k: dict(v, size=size) for k, v in _cases(size).items()}
    names = list(cases)
    pairs = [_pair(size, seed=i) for i in range(len(names))]
    imgs = torch.from_numpy(np.stack([p[0] for p in pairs])).cuda()
    masks = torch.from_numpy(np.stack([p[1] for p in pairs])).cuda()
    x, m, fields, _ = augment_batch(imgs, masks, None, params=[cases[k] for k in names], want_fields=True)
    torch.cuda.synchronize()
    x, m, fields = x.cpu().numpy()[:, 0], m.cpu().numpy(), fields.cpu().numpy()
    for i, name in enumerate(names):
        ref_i, ref_m = A.apply_params(pairs[i][0], pairs[i][1], cases[name], fields=(fields[i, 0], fields[i, 1]))
        got_u8 = np.rint((x[i] * np.float32(0.226) + np.float32(0.449)) * 255).astype(np.int64)
        d = np.abs(got_u8 - ref_i.astype(np.int64))
        exact = name in ("identity", "crop_full", "flip", "rot1", "rot3_t", "flip_rot2_t", "bc", "gamma")
        assert (m[i] != ref_m).mean() <= (0 if exact else 2e-3), (name, (m[i] != ref_m).mean())
        assert d.max() <= (0 if exact else 2) and (d > 0).mean() <= (0 if exact else 0.02), (name, int(d.max()), (d > 0).mean())
        ref_x = ((ref_i.astype(np.float32) / 255) - np.float32(0.449)) / np.float32(0.226)
        if exact:
            assert np.array_equal(x[i].view(np.uint32), ref_x.view(np.uint32)), name       # the normalisation is NumPy's, bit for bit
    # the displacement fields look like scipy's: smooth, zero-mean, a few pixels of amplitude, different per field
    e = names.index("elastic")
    ref_dx, _ = A.elastic_fields(size, size, 1)
    assert abs(fields[e].mean()) < 1.5 and 0.5 < fields[e, 0].std() / ref_dx.std() < 2.0 and not np.allclose(fields[e, 0], fields[e, 1])
    assert np.abs(np.diff(fields[e, 0], axis=1)).max() < 1.0


@pytest.mark.gpu
def test_device_augmentation_feeds_training_batches():
    """prepare_training_batch on a raw uint8 batch: random reference-pipeline draws per sample, normalised fp32 input, one-hot targets."""
    from volume_segmantics_amd.utilities import base_data_utils as utils
    pairs = [_pair(64, seed=i) for i in range(8)]
    batch = [torch.from_numpy(np.stack([p[0] for p in pairs]))[:, None], torch.from_numpy(np.stack([np.minimum(p[1], 1) for p in pairs]))]
    x0, t0 = utils.prepare_training_batch(batch, "cuda:0", 2)                          # no rng: normalise only
    ref = ((np.stack([p[0] for p in pairs]).astype(np.float32) / 255) - np.float32(0.449)) / np.float32(0.226)
    assert np.array_equal(x0.cpu().numpy()[:, 0], ref) and t0.shape == (8, 2, 64, 64) and t0.dtype == torch.uint8
    x1, t1 = utils.prepare_training_batch(batch, "cuda:0", 2, augment_rng=np.random.default_rng(0))
    assert x1.shape == x0.shape and torch.isfinite(x1).all() and not torch.equal(x1, x0)
    assert (t1.sum(1) == 1).all()
