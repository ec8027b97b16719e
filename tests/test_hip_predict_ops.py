"""-m gpu: prediction-side kernels through the C ABI - bit-exact against the numpy oracle
(integer / byte / index work) and the reference-generated goldens."""
import numpy as np
import pytest
import torch

from hip_helpers import DEV, dirmap_from_view, lib, sync
from oracle import predictor_numpy as P

pytestmark = pytest.mark.gpu


def test_slices_gather_bit_exact_all_12_directions(golden):
    L = lib()
    vol = golden("g3_predict_29x64x40_c4.npz")["vol"]
    vd = torch.from_numpy(vol).to(DEV)
    for view in P.direction_views(vol, 12):
        m = dirmap_from_view(L, vol, view)
        x = torch.full((view.shape[0], m.hp, m.wp), float("nan"), device=DEV)
        for s0 in range(0, view.shape[0], 5):  # ragged batches
            nb = min(5, view.shape[0] - s0)
            L.check(L.lib.vs_slices_gather(L.ptr(vd), m, s0, nb, L.ptr(x[s0:]), None))
        sync()
        ref = np.stack([P.preprocess_slice(view[i]) for i in range(view.shape[0])])
        assert ref.dtype == np.float32
        assert np.array_equal(x.cpu().numpy().view(np.uint32), ref.view(np.uint32))  # bit-exact fp32


def test_typed_slices_gather_equals_reference_dataset(golden):
    """Prediction volumes that are not uint8 (clip_data: False): vs_slices_gather_typed against g9, the network input the
    reference's own VolSeg2dPredictionDataset builds - bit-exact (float64 volumes: the reference's float64 input rounded to
    float32, which is what a float32 network can take)."""
    L = lib()
    g = golden("g9_prediction_inputs_typed.npz")
    for name in ("u8", "u16", "i16", "i32", "u32", "i64", "f32", "f64"):
        vol, ref = np.ascontiguousarray(g[name + "__in"]), g[name + "__x"][:, 0]
        m = dirmap_from_view(L, vol, vol)
        raw = torch.from_numpy(vol.reshape(-1).view(np.uint8).copy()).to(DEV)
        x = torch.full((vol.shape[0], m.hp, m.wp), float("nan"), device=DEV)
        L.check(L.lib.vs_slices_gather_typed(L.VS_VOL[vol.dtype.name], L.ptr(raw), m, 0, vol.shape[0], L.ptr(x), None))
        sync()
        assert np.array_equal(x.cpu().numpy().view(np.uint32), ref.astype(np.float32).view(np.uint32)), name
        # a transposed direction of the same volume (strided reads of a wider type)
        view = vol.swapaxes(0, 2)
        mv = dirmap_from_view(L, vol, view)
        xv = torch.empty((view.shape[0], mv.hp, mv.wp), device=DEV)
        L.check(L.lib.vs_slices_gather_typed(L.VS_VOL[vol.dtype.name], L.ptr(raw), mv, 0, view.shape[0], L.ptr(xv), None))
        sync()
        refv = np.stack([P.preprocess_slice(view[i]) for i in range(view.shape[0])]).astype(np.float32)
        assert np.array_equal(xv.cpu().numpy().view(np.uint32), refv.view(np.uint32)), name


def test_reflect101_when_padding_exceeds_size():
    L = lib()
    vol = np.random.default_rng(0).integers(0, 256, size=(3, 10, 13), dtype=np.uint8)  # pads 11 / 9 > size-1? (10->32)
    m = dirmap_from_view(L, vol, vol)
    x = torch.empty((3, m.hp, m.wp), device=DEV)
    vd = torch.from_numpy(vol).to(DEV)
    L.check(L.lib.vs_slices_gather(L.ptr(vd), m, 0, 3, L.ptr(x), None))
    sync()
    ref = np.stack([P.preprocess_slice(vol[i]) for i in range(3)])
    assert np.array_equal(x.cpu().numpy(), ref)


@pytest.mark.parametrize("classes", [2, 4])
def test_logits_to_volume_modes(classes):
    L = lib()
    rng = np.random.default_rng(3)
    vol = np.zeros((7, 29, 40), np.uint8)
    nvox = vol.size
    views = P.direction_views(vol, 12)
    keys = torch.zeros(nvox, dtype=torch.int32, device=DEV)
    votes = torch.zeros((classes, nvox), dtype=torch.uint8, device=DEV)
    ref_key = np.zeros(vol.shape, np.uint32)
    ref_votes = np.zeros((classes, *vol.shape), np.uint8)
    for d, view in enumerate(views):
        m = dirmap_from_view(L, vol, view)
        depth = view.shape[0]
        logits = torch.from_numpy(rng.standard_normal((depth, classes, m.hp, m.wp)).astype(np.float32) * 3)
        logits[:, :, ::3, ::2] = 1.0  # exact class ties -> first index must win
        probs = torch.softmax(logits, 1)
        lab = torch.argmax(probs, 1)[:, m.crop_top:m.crop_top + m.h, m.crop_left:m.crop_left + m.w].numpy().astype(np.uint8)
        mp = torch.gather(probs, 1, torch.argmax(probs, 1, keepdim=True)).squeeze(1)
        mp = mp[:, m.crop_top:m.crop_top + m.h, m.crop_left:m.crop_left + m.w].numpy().astype(np.float16)
        ld = logits.to(DEV)
        labels = torch.full((nvox,), 255, dtype=torch.uint8, device=DEV)
        probs16 = torch.zeros(nvox, dtype=torch.float16, device=DEV)
        L.check(L.lib.vs_logits_to_volume(L.ptr(ld), classes, m, 0, depth, 0, d, L.ptr(labels), L.ptr(probs16), None, None, nvox, None))
        L.check(L.lib.vs_logits_to_volume(L.ptr(ld), classes, m, 0, depth, 1, d, None, None, L.ptr(keys), None, nvox, None))
        L.check(L.lib.vs_logits_to_volume(L.ptr(ld), classes, m, 0, depth, 2, d, None, None, None, L.ptr(votes), nvox, None))
        sync()
        # scatter the reference through the same numpy view
        ref_l = np.zeros(vol.shape, np.uint8); ref_p = np.zeros(vol.shape, np.float16)
        lv, pv = [P.direction_views(a, 12)[d] for a in (ref_l, ref_p)]
        lv[...] = lab; pv[...] = mp
        got_l = labels.cpu().numpy().reshape(vol.shape)
        got_p = probs16.cpu().numpy().reshape(vol.shape)
        assert np.array_equal(got_l, ref_l)
        ulp = np.abs(got_p.view(np.int16).astype(np.int32) - ref_p.view(np.int16).astype(np.int32))
        assert ulp.max() <= 1  # expf on device vs torch's vectorised exp: <= 1 fp16 ulp
        ref_key = np.maximum(ref_key, P.pack_key(got_p, got_l, d))
        ref_votes += P.one_hot_encode_array(got_l, classes)
    assert np.array_equal(keys.cpu().numpy().view(np.uint32).reshape(vol.shape), ref_key)
    assert np.array_equal(votes.cpu().numpy().reshape(ref_votes.shape), ref_votes)
    lab_u = torch.empty(nvox, dtype=torch.uint8, device=DEV); pr_u = torch.empty(nvox, dtype=torch.float16, device=DEV)
    L.check(L.lib.vs_keys_unpack(L.ptr(keys), L.ptr(lab_u), L.ptr(pr_u), nvox, None))
    sync()
    kl, kp = P.unpack_key(ref_key)
    assert np.array_equal(lab_u.cpu().numpy().reshape(vol.shape), kl)
    assert np.array_equal(pr_u.cpu().numpy().reshape(vol.shape).view(np.uint16), kp.view(np.uint16))


def test_merge_maxprob_matches_reference_chain_bit_exact(golden):
    L = lib()
    g = golden("g4_merge_ties.npz")
    dl, dp = g["dlabels"], g["dprobs"]
    l0 = torch.from_numpy(dl[0].copy()).to(DEV); p0 = torch.from_numpy(dp[0].copy()).to(DEV)
    for d in range(1, dl.shape[0]):
        l1 = torch.from_numpy(dl[d].copy()).to(DEV); p1 = torch.from_numpy(dp[d].copy()).to(DEV)
        L.check(L.lib.vs_merge_maxprob(L.ptr(l0), L.ptr(p0), L.ptr(l1), L.ptr(p1), l0.numel(), None))
        sync()
        assert np.array_equal(l0.cpu().numpy(), g["chain_labels"][d - 1])
        assert np.array_equal(p0.cpu().numpy().view(np.uint16), g["chain_probs"][d - 1].view(np.uint16))


def test_merge_large_random_vs_numpy_and_empty():
    L = lib()
    rng = np.random.default_rng(5)
    n = 3_000_017  # ragged size
    lab = rng.integers(0, 4, size=(2, n), dtype=np.uint8)
    prob = rng.random((2, n)).astype(np.float16)
    prob[:, ::7] = np.float16(1.0)  # ties at p = 1.0 are the common case in practice
    l0, p0 = torch.from_numpy(lab[0].copy()).to(DEV), torch.from_numpy(prob[0].copy()).to(DEV)
    l1, p1 = torch.from_numpy(lab[1].copy()).to(DEV), torch.from_numpy(prob[1].copy()).to(DEV)
    L.check(L.lib.vs_merge_maxprob(L.ptr(l0), L.ptr(p0), L.ptr(l1), L.ptr(p1), n, None))
    L.check(L.lib.vs_merge_maxprob(L.ptr(l0), L.ptr(p0), L.ptr(l1), L.ptr(p1), 0, None))
    sync()
    P.merge_vols_in_mem(prob, lab)
    assert np.array_equal(l0.cpu().numpy(), lab[0]) and np.array_equal(p0.cpu().numpy(), prob[0])
