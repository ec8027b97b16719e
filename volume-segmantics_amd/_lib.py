"""ctypes binding of libvolseg_hip.so (declared in include/volseg_hip.h).

The product path has no CPU fallback: if the shared library is missing this module raises at
import, and every wrapper raises RuntimeError(vs_last_error()) on a non-zero status."""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

_HERE = Path(__file__).resolve().parent
LIB_PATH = Path(os.environ.get("VOLSEG_HIP_LIB", _HERE / "lib" / "libvolseg_hip.so"))

VS_F32, VS_BF16, VS_F16 = 0, 1, 2
VS_VOL = {"float32": 0, "float64": 1, "uint8": 2, "int8": 3, "uint16": 4, "int16": 5, "uint32": 6, "int32": 7,
          "int64": 8, "uint64": 9}   # volume dtypes


class VolsegHipMissing(ImportError):
    pass


if not LIB_PATH.exists():
    raise VolsegHipMissing(
        f"{LIB_PATH} not found - build it with `python -c 'import __graft_entry__ as g; g.build()'` "
        f"(or `make -C {_HERE / 'csrc'}`); there is no CPU fallback for the HIP path")

# torch first: PyTorch-ROCm ships its own HIP runtime, and the library's dependency on libamdhip64 must resolve to THAT copy - loaded
# the other way round (this module before torch, e.g. build() and smoke() in one process) the process holds two runtimes and the
# second one finds "no ROCm-capable device"
import torch  # noqa: E402,F401

lib = C.CDLL(str(LIB_PATH))


class ConvDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "dtype", "n", "hin", "win", "c0", "c1", "up0", "cout", "kh", "kw", "stride", "pad", "relu", "out_f32", "split_c", "groups", "dilation")]


class ConvTrain(C.Structure):
    """vs_conv_train: the training forms of a convolution launch (statistics epilogue, pooled / masked data gradients, normalise on load)"""
    _fields_ = [("stats_bins", C.c_void_p), ("stats_nb", C.c_int32), ("stats_partial", C.c_void_p), ("pool0", C.c_int32),
                ("bz", C.c_void_p), ("by", C.c_void_p), ("bmean", C.c_void_p), ("binvstd", C.c_void_p), ("bgamma", C.c_void_p),
                ("bbeta", C.c_void_p), ("bstats_partial", C.c_void_p), ("brelu", C.c_int32),
                ("nl_bins", C.c_void_p), ("nl_nb", C.c_int32), ("nl_rows", C.c_int64), ("nl_eps", C.c_float), ("nl_mom", C.c_float),
                ("nl_mean", C.c_void_p), ("nl_invstd", C.c_void_p), ("nl_rm", C.c_void_p), ("nl_rv", C.c_void_p),
                ("nl_gamma", C.c_void_p), ("nl_beta", C.c_void_p), ("nl_y", C.c_void_p)]


class DirMap(C.Structure):
    _fields_ = [("base", C.c_int64), ("ss", C.c_int64), ("sh", C.c_int64), ("sw", C.c_int64)] + [
        (n, C.c_int32) for n in ("depth", "h", "w", "hp", "wp", "pad_top", "pad_left", "crop_top", "crop_left")]


class AdamwArgs(C.Structure):
    _fields_ = [("params", C.c_void_p), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p), ("lr", C.c_float),
                ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float), ("weight_decay", C.c_float), ("step", C.c_int32),
                ("hyper", C.c_void_p)]


class AugParams(C.Structure):
    _fields_ = ([(n, C.c_int32) for n in ("crop", "y1", "x1", "ch", "cw", "flip_v", "rot_k", "transpose", "distort")] +
                [("inv_affine", C.c_float * 6), ("k", C.c_float), ("cx", C.c_float), ("cy", C.c_float), ("noise_seed", C.c_uint32),
                 ("clahe_clip", C.c_float), ("clahe_limit", C.c_int32)])


P, I, I64, F, SZ = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_size_t
_SIGS = {
    "vs_last_error": (C.c_char_p, []),
    "vs_version": (I, []),
    "vs_conv2d_fwd": (I, [C.POINTER(ConvDesc), P, P, P, P, P, P, P, P, P]),
    "vs_conv2d_variant": (I, [C.POINTER(ConvDesc)]),
    "vs_conv2d_train": (I, [C.POINTER(ConvDesc), P, P, P, P, P, P, C.POINTER(ConvTrain), P]),
    "vs_conv2d_train_variant": (I, [C.POINTER(ConvDesc), C.POINTER(ConvTrain)]),
    "vs_conv2d_stat_rows": (I, [C.POINTER(ConvDesc), C.POINTER(ConvTrain)]),
    "vs_stat_scale": (C.c_double, [I]),
    "vs_conv2d_pair_ok": (I, [C.POINTER(ConvDesc), C.POINTER(ConvDesc)]),
    "vs_conv2d_pair_fwd": (I, [C.POINTER(ConvDesc), C.POINTER(ConvDesc), P, P, P, P, P, P, P, P, P]),
    "vs_conv2d_wgrad_workspace": (SZ, [C.POINTER(ConvDesc)]),
    "vs_conv2d_wgrad": (I, [C.POINTER(ConvDesc), P, P, P, P, P, SZ, P]),
    "vs_head_dgrad_planes": (I, [I, P, P, P, I, I, I, I, I, P]),
    "vs_head_wgrad_planes_workspace": (SZ, [I, I, I, I, I]),
    "vs_head_wgrad_planes": (I, [I, P, P, P, P, SZ, I, I, I, I, P]),
    "vs_weights_prepare": (I, [I, P, P, P, I, I, I, P]),
    "vs_weights_prepare_grouped": (I, [I, P, P, P, I, I, I, P]),
    "vs_depth_to_space2": (I, [I, P, P, I, I, I, I, P, P, P, I, P]),
    "vs_space_to_depth2": (I, [I, P, P, I, I, I, I, P]),
    "vs_convt_weights_prepare": (I, [I, P, P, P, I, I, P]),
    "vs_convt_wgrad_gather": (I, [P, P, I, I, P]),
    "vs_upsample2x_add": (I, [I, P, P, P, I, I, I, I, P]),
    "vs_gn_workspace": (SZ, [I, I]),
    "vs_gn_bwd_workspace": (SZ, [I, I, I]),
    "vs_gn_fwd": (I, [I, P, P, P, I, P, P, I, C.c_int64, I, I, C.c_float, P, SZ, P]),
    "vs_gn_bwd": (I, [I, P, P, P, P, P, I, P, P, P, I, C.c_int64, I, I, P, SZ, P]),
    "vs_bilinear_up": (I, [I, P, P, I, I, I, I, I, P]),
    "vs_bilinear_up_bwd": (I, [I, P, P, I, I, I, I, I, I, P]),
    "vs_bilinear_up_planes": (I, [P, P, I, I, I, I, P]),
    "vs_bilinear_up_planes_bwd": (I, [P, P, I, I, I, I, P]),
    "vs_dropout2d_mask": (I, [P, I, I, C.c_float, C.c_uint32, P, C.c_int64, P]),
    "vs_channel_scale": (I, [I, P, P, P, I, C.c_int64, I, P]),
    "vs_dwconv3x3": (I, [I, P, P, P, I, I, I, I, I, I, P]),
    "vs_dwconv3x3_wgrad_workspace": (SZ, [I]),
    "vs_dwconv3x3_wgrad": (I, [I, P, P, P, I, I, I, I, I, P, SZ, P]),
    "vs_spatial_sum": (I, [I, P, P, I, C.c_int64, I, C.c_float, P]),
    "vs_broadcast_rows": (I, [I, P, P, I, C.c_int64, I, C.c_float, I, P]),
    "vs_dropout": (I, [I, P, P, C.c_int64, C.c_float, C.c_uint32, P, C.c_int64, P]),
    "vs_dilated_im2col": (I, [I, P, P, I, I, I, I, I, I, I, P]),
    "vs_maxpool2x2": (I, [I, P, P, I, I, I, I, P]),
    "vs_maxpool2x2_bwd": (I, [I, P, P, P, I, I, I, I, I, P]),
    "vs_conv_to_plane": (I, [I, P, P, P, P, I, I, I, I, I, P]),
    "vs_conv_to_plane_bwd": (I, [I, P, P, P, P, P, P, I, I, I, I, I, P]),
    "vs_fpa_arena_floats": (SZ, [I, I, I]),
    "vs_fpa_dz1_offset": (SZ, [I, I, I]),
    "vs_fpa_pyramid_fwd": (I, [P, P, P, I, I, I, I, P]),
    "vs_fpa_pyramid_bwd": (I, [P, P, P, P, I, I, I, P]),
    "vs_fpa_combine": (I, [I, P, P, P, P, I, C.c_int64, I, P]),
    "vs_fpa_combine_bwd": (I, [I, P, P, P, P, P, I, C.c_int64, I, P]),
    "vs_sigmoid": (I, [I, P, P, C.c_int64, P]),
    "vs_sigmoid_bwd": (I, [I, P, P, P, C.c_int64, P]),
    "vs_bn_fold_bias": (I, [P, P, P, I, P]),
    "vs_pab_attention_fwd": (I, [I, P, P, P, P, P, P, P, I, I, I, I, P]),
    "vs_pab_attention_bwd": (I, [I, P, P, P, P, P, P, P, P, P, I, I, I, I, P]),
    "vs_pab_scratch_bytes": (SZ, [I, I, I]),
    "vs_bn2_workspace": (SZ, [I]),
    "vs_bn2_stats": (I, [I, P, I64, I, F, F, P, P, P, P, P, SZ, P]),
    "vs_bn2_apply": (I, [I, P, P, P, P, P, I, F, P, I64, I, P]),
    "vs_bn2_bwd": (I, [I, P, P, P, P, P, P, I, P, P, P, I64, I, P, SZ, P]),
    "vs_dwconv2d": (I, [I, P, P, P, I, I, I, I, I, I, I, I, I, I, I, P]),
    "vs_dwconv2d_bwd_data": (I, [I, P, P, P, I, I, I, I, I, I, I, I, I, I, I, P]),
    "vs_dwconv2d_affine": (I, [I, P, P, P, P, I, P, I, I, I, I, I, I, I, I, I, I, P]),
    "vs_dwconv2d_wgrad_workspace": (SZ, [I, I]),
    "vs_dwconv2d_wgrad": (I, [I, P, P, P, I, I, I, I, I, I, I, I, I, I, I, P, SZ, P]),
    "vs_sample_scale_add": (I, [I, P, P, P, P, I, I64, P]),
    "vs_sample_rowsum": (I, [I, P, P, P, I, I64, I, F, P]),
    "vs_sample_rowsum_workspace": (SZ, [I, I]),
    "vs_sample_rowsum_ws": (I, [I, P, P, P, I, I64, I, F, P, SZ, P]),
    "vs_radix2_softmax": (I, [I, P, P, I, I, P]),
    "vs_radix2_softmax_bwd": (I, [I, P, P, P, I, I, P]),
    "vs_radix2_gated_sum": (I, [I, P, P, P, I, I64, I, P]),
    "vs_radix2_gated_sum_bwd": (I, [I, P, P, P, I, I64, I, P]),
    "vs_sample_rowsum_b": (I, [I, P, P, I, P, I, I64, I, P, SZ, P]),
    "vs_se_gate_fwd": (I, [I, P, P, P, P, P, P, P, I, I, I, I, P]),
    "vs_se_gate_bwd": (I, [I, P, P, P, P, P, P, P, P, P, P, P, P, I, I, I, I, P]),
    "vs_se_gate_scratch_floats": (SZ, [I, I, I]),
    "vs_channel_gate": (I, [I, P, P, P, I, C.c_int64, I, P]),
    "vs_channel_dot": (I, [I, P, P, P, I, C.c_int64, I, P]),
    "vs_colsum_workspace": (SZ, [I]),
    "vs_colsum": (I, [I, P, C.c_int64, I, P, P, SZ, P]),
    "vs_stem_fwd": (I, [I, P, P, P, P, I, P, I, I, I, P]),
    "vs_stem_wgrad": (I, [I, P, P, P, P, SZ, I, I, I, P]),
    "vs_stem_wgrad_workspace": (SZ, [I, I, I]),
    "vs_bn_stats": (I, [I, P, I64, I, F, F, P, P, P, P, P, SZ, P]),
    "vs_bn_workspace": (SZ, [I64, I]),
    "vs_bn_apply": (I, [I, P, P, P, P, P, P, I, P, I64, I, P]),
    "vs_bn_bwd": (I, [I, P, P, P, P, P, P, I, P, P, P, P, I64, I, P, SZ, P]),
    "vs_bn_bwd_recompute": (I, [I, P, P, P, P, P, P, P, I, P, P, P, P, I64, I, P, SZ, P]),
    "vs_bn_fold": (I, [P, P, P, P, F, P, P, I, P]),
    "vs_maxpool_fwd": (I, [I, P, P, P, I, I, I, I, P]),
    "vs_maxpool_bwd": (I, [I, P, P, P, I, I, I, I, I, P]),
    "vs_upsample2x_bwd": (I, [I, P, P, I, I, I, I, P]),
    "vs_zero_stuff2x": (I, [I, P, P, I, I, I, I, P]),
    "vs_channel_slice": (I, [I, P, I, I, P, I, I, I, I64, I, P]),
    "vs_unet_num_tensors": (I, [I]),
    "vs_unet_tensor_info": (I, [I, I, C.c_char_p, I, C.POINTER(I64), C.POINTER(I), C.POINTER(I), C.POINTER(I64)]),
    "vs_unet_param_elems": (I64, [I]),
    "vs_unet_bnstate_elems": (I64, [I]),
    "vs_unet_create": (I, [C.POINTER(P), I, I, I, I, I]),
    "vs_unet_create_ex": (I, [C.POINTER(P), I, I, I, I, I, I]),
    "vs_unet_num_tensors_ex": (I, [I, I]),
    "vs_unet_tensor_info_ex": (I, [I, I, I, C.c_char_p, I, C.POINTER(I64), C.POINTER(I), C.POINTER(I), C.POINTER(I64)]),
    "vs_unet_param_elems_ex": (I64, [I, I]),
    "vs_unet_bnstate_elems_ex": (I64, [I, I]),
    "vs_unet_destroy": (None, [P]),
    "vs_unet_workspace_bytes": (SZ, [P, I]),
    "vs_unet_prepare": (I, [P, P, P, I, P, P]),
    "vs_unet_forward": (I, [P, P, P, P, I, I, P, P, P]),
    "vs_unet_backward": (I, [P, P, P, P, I, I, P, P, P]),
    "vs_unet_backward_range": (I, [P, P, P, P, I, I, P, P, P, I, I]),
    "vs_unet_backward_adamw": (I, [P, P, P, I, I, P, P, P, C.POINTER(AdamwArgs)]),
    "vs_unet_backward_adamw_part": (I, [P, P, P, I, I, P, P, P, C.POINTER(AdamwArgs), I, I, I]),
    "vs_unet_backward_part": (I, [P, P, P, P, I, I, P, P, P, I, I, I]),
    "vs_unet_set_rng": (I, [P, C.c_uint32, P]),
    "vs_debug_mfma_rate": (I, [I, I, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "vs_comm_unique_id": (I, [P]),
    "vs_comm_init": (I, [C.POINTER(P), I, I, P]),
    "vs_comm_destroy": (None, [P]),
    "vs_comm_size": (I, [P]),
    "vs_comm_rank": (I, [P]),
    "vs_comm_allreduce_sum_f32": (I, [P, P, C.c_int64, P]),
    "vs_comm_allreduce_max_u32": (I, [P, P, C.c_int64, P]),
    "vs_comm_reduce_scatter_max_u32": (I, [P, P, P, C.c_int64, P]),
    "vs_comm_allgather": (I, [P, P, P, C.c_int64, P]),
    "vs_comm_broadcast": (I, [P, P, C.c_int64, I, P]),
    "vs_unet_dropout_mask_offset": (C.c_int64, [P]),
    "vs_unet_drop_connect_masks": (I, [P, P, P, P, I]),
    "vs_unet_adamw_range": (I, [P, I, P, P, P, C.POINTER(AdamwArgs), I, I]),
    "vs_unet_prepare_range": (I, [P, P, P, P, I, I]),
    "vs_unet_flip_weight_set": (I, [P]),
    "vs_unet_side_stream_overlaps": (I, [P, P, C.POINTER(I)]),
    "vs_unet_side_stream": (I, [P, P, I, C.POINTER(P)]),
    "vs_unet_weight_set": (I, [P]),
    "vs_capture_begin": (P, []),
    "vs_capture_end": (I, [C.POINTER(P)]),
    "vs_capture_abort": (I, []),
    "vs_graph_launch": (I, [P, P]),
    "vs_graph_num_nodes": (I64, [P]),
    "vs_graph_destroy": (None, [P]),
    "vs_train_hyper_set": (I, [P, F, F, F, F, F, I, P, I, P]),
    "vs_unet_unit_param_offset": (I64, [P, I]),
    "vs_unet_num_units": (I, [P]),
    "vs_unet_nl_plan": (I, [P, I, C.POINTER(I), I]),
    "vs_unet_set_stats_hook": (I, [P, P, P, I]),
    "vs_unet_debug_unit": (I, [P, I, C.c_char_p, I, C.POINTER(I), C.POINTER(I), C.POINTER(I), C.POINTER(SZ), C.POINTER(SZ),
                               C.POINTER(SZ), C.POINTER(SZ)]),
    "vs_profile_enable": (I, [I]),
    "vs_profile_enabled": (I, []),
    "vs_profile_num_kinds": (I, []),
    "vs_profile_kind_name": (C.c_char_p, [I]),
    "vs_profile_read": (I, [C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(I64)]),
    "vs_profile_read_raw": (I, [I, C.POINTER(I), C.POINTER(I), C.POINTER(I), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "vs_set_option": (I, [C.c_char_p, I]),
    "vs_get_option": (I, [C.c_char_p]),
    "vs_debug_probe": (I, [P, SZ]),
    "vs_adamw_step": (I, [P, P, P, P, P, I64, F, F, F, F, F, I, P]),
    "vs_dice_workspace": (SZ, [I]),
    "vs_dice_loss_fwd": (I, [P, P, I, I, I, I64, F, P, P, SZ, P]),
    "vs_dice_loss_bwd": (I, [P, P, I, P, I, I, I64, F, P, P, P]),
    "vs_seg_loss_workspace": (SZ, [I]),
    "vs_seg_loss_fwd": (I, [I, P, P, I, I, I, I64, F, F, F, P, P, SZ, P]),
    "vs_seg_loss_bwd": (I, [I, P, P, I, P, I, I, I64, P, P, P]),
    "vs_mean_iou_workspace": (SZ, [I, I]),
    "vs_mean_iou": (I, [P, P, I, I, I, I, I64, P, P, SZ, P]),
    "vs_onehot_u8": (I, [P, I, I, I64, P, P]),
    "vs_slices_gather": (I, [P, C.POINTER(DirMap), I, I, P, P]),
    "vs_slices_gather_typed": (I, [I, P, C.POINTER(DirMap), I, I, P, P]),
    "vs_logits_to_volume": (I, [P, I, C.POINTER(DirMap), I, I, I, I, P, P, P, P, I64, P]),
    "vs_unet_forward_to_volume": (I, [P, P, P, P, I, P, P, C.POINTER(DirMap), I, I, I, P, P, P, P, I64]),
    "vs_merge_maxprob": (I, [P, P, P, P, I64, P]),
    "vs_keys_unpack": (I, [P, P, P, I64, P]),
    "vs_augment_workspace": (SZ, [I, I]),
    "vs_augment_batch": (I, [P, P, I, I, P, P, P, P, P, P, SZ, P, P]),
    "vs_volume_sum_workspace": (SZ, [I64]),
    "vs_volume_sum": (I, [I, P, I64, I, C.c_double, P, SZ, P, P]),
    "vs_clip_to_uint8": (I, [I, P, I64, C.c_double, C.c_double, C.c_double, P, P, P]),
    "vs_downsample2x_mean": (I, [I, P, P, I, I, I, P]),
}
for _name, (_res, _args) in _SIGS.items():
    _fn = getattr(lib, _name)  # AttributeError here = header and library out of sync
    _fn.restype, _fn.argtypes = _res, _args


def last_error() -> str:
    return lib.vs_last_error().decode()


def check(rc: int) -> None:
    if rc != 0:
        raise RuntimeError(f"libvolseg_hip error {rc}: {last_error()}")


def ptr(t) -> int | None:
    """Device/host address of a torch tensor (None passes a null pointer)."""
    return None if t is None else t.data_ptr()


def stream_ptr() -> int:
    import torch

    return torch.cuda.current_stream().cuda_stream


def dtype_code(torch_dtype) -> int:
    import torch

    if torch_dtype == torch.float32:
        return VS_F32
    if torch_dtype == torch.bfloat16:
        return VS_BF16
    if torch_dtype == torch.float16:
        return VS_F16
    raise ValueError(f"unsupported compute dtype {torch_dtype}")


def unet_tensor_table(classes: int, encoder: int = 34):
    """[(name, shape, kind, offset)] in smp state_dict order (see vs_unet_tensor_info)."""
    n = lib.vs_unet_num_tensors_ex(classes, encoder)
    if n < 0:
        raise RuntimeError(last_error())
    out = []
    name = C.create_string_buffer(128)
    shape = (I64 * 4)()
    ndim, kind, off = I(), I(), I64()
    for i in range(n):
        check(lib.vs_unet_tensor_info_ex(classes, encoder, i, name, 128, shape, C.byref(ndim), C.byref(kind), C.byref(off)))
        out.append((name.value.decode(), tuple(shape[: ndim.value]), kind.value, off.value))
    return out


def profile_read():
    """{kind: dict(ms, flops, bytes, calls)} accumulated since vs_profile_enable(1)."""
    n = lib.vs_profile_num_kinds()
    ms, fl, by = (C.c_double * n)(), (C.c_double * n)(), (C.c_double * n)()
    calls = (I64 * n)()
    check(lib.vs_profile_read(ms, fl, by, calls))
    return {lib.vs_profile_kind_name(k).decode(): dict(ms=ms[k], flops=fl[k], bytes=by[k], calls=calls[k]) for k in range(n)}


def profile_read_raw(max_n: int = 1 << 16):
    """[(kind_name, unit_index, ms, flops, bytes, variant)] in launch order."""
    kind, tag, var = (I * max_n)(), (I * max_n)(), (I * max_n)()
    ms, fl, by = (C.c_double * max_n)(), (C.c_double * max_n)(), (C.c_double * max_n)()
    n = lib.vs_profile_read_raw(max_n, kind, tag, var, ms, fl, by)
    if n < 0:
        raise RuntimeError("vs_profile_read_raw failed")
    return [(lib.vs_profile_kind_name(kind[i]).decode(), tag[i], ms[i], fl[i], by[i], var[i]) for i in range(n)]


# SyncBatchNorm hook (include/volseg_hip.h: vs_stats_hook_t): int (*)(void* user, void* values, int64_t count, int kind, void* stream)
STATS_HOOK = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p)


def set_option(name: str, value: int) -> None:
    check(lib.vs_set_option(name.encode(), int(value)))


def unit_names(handle) -> list[str]:
    """weight-tensor name of every unit of a network plan (index = profiler tag)."""
    out = []
    name = C.create_string_buffer(128)
    c, h, w = I(), I(), I()
    o = [SZ() for _ in range(4)]
    for u in range(lib.vs_unet_num_units(handle)):
        check(lib.vs_unet_debug_unit(handle, u, name, 128, C.byref(c), C.byref(h), C.byref(w), *[C.byref(x) for x in o]))
        out.append(f"{name.value.decode()} [{c.value}x{h.value}x{w.value}]")
    return out
