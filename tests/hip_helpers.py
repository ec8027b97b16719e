"""Helpers for the -m gpu parity tests: everything goes through the C ABI (ctypes)."""
import ctypes as C

import numpy as np
import torch

DEV = "cuda:0"


def lib():
    from volume_segmantics_amd import _lib
    return _lib


def tdtype(code):
    return {0: torch.float32, 1: torch.bfloat16, 2: torch.float16}[code]     # VS_F32 / VS_BF16 / VS_F16 (inference only)


def to_nhwc(x_nchw, code):
    """(N,C,H,W) fp32 cpu -> (N,H,W,C) device tensor in the compute dtype."""
    return x_nchw.permute(0, 2, 3, 1).contiguous().to(DEV, tdtype(code))


def from_nhwc(y):
    return y.float().cpu().permute(0, 3, 1, 2).contiguous()


def w_krsc(w_oihw, code):
    return w_oihw.permute(0, 2, 3, 1).contiguous().to(DEV, tdtype(code))


def rounded(x, code):
    """Value the device sees for a CPU fp32 tensor (bf16 rounding for the bf16 path)."""
    return x if code == 0 else x.to(tdtype(code)).float()


def tol(code, scale=1.0):
    if code == 2:       # fp16: 11 significant bits (8x bf16's): inputs rounded beforehand, fp32 accumulation, one rounding at the store
        return dict(rtol=2e-3, atol=2e-3 * scale)
    return dict(rtol=2e-4, atol=2e-4 * scale) if code == 0 else dict(rtol=2e-2, atol=2e-2 * scale)


def sync():
    torch.cuda.synchronize()


def conv_desc(L, code, n, hin, win, c0, cout, k, stride, pad, c1=0, up0=0, relu=0, out_f32=0, split_c=0, groups=0, dilation=0):
    return L.ConvDesc(dtype=code, n=n, hin=hin, win=win, c0=c0, c1=c1, up0=up0, cout=cout, kh=k, kw=k,
                      stride=stride, pad=pad, relu=relu, out_f32=out_f32, split_c=split_c, groups=groups, dilation=dilation)


def dirmap_from_view(L, vol, view):
    """vs_dirmap for a numpy *view* (rot90/swapaxes of vol) - strides read off numpy itself."""
    from oracle import predictor_numpy as P
    item = vol.itemsize
    base = (view.__array_interface__["data"][0] - vol.__array_interface__["data"][0]) // item
    d, h, w = view.shape
    hp, wp = P.get_padded_dimension(h), P.get_padded_dimension(w)
    return L.DirMap(base=base, ss=view.strides[0] // item, sh=view.strides[1] // item, sw=view.strides[2] // item,
                    depth=d, h=h, w=w, hp=hp, wp=wp, pad_top=P.pad_offsets(h)[0], pad_left=P.pad_offsets(w)[0],
                    crop_top=P.crop_offset(hp, h), crop_left=P.crop_offset(wp, w))
