"""The drop-in seam: ``create_model_on_device`` / ``create_model_from_file``
(volume_segmantics/model/model_2d.py:10-57).  For every model type of the reference (U_NET / U_NET_PLUS_PLUS / LINKNET / FPN / DEEPLABV3 / DEEPLABV3_PLUS / MA_NET / PAN) over
the encoders of its list (resnet34 / resnet50 / resnext50_32x4d / efficientnet-b3 / -b4 / timm-resnest50d / -101e, + resnet18) the
returned ``nn.Module`` is the HIP engine (engine.VolSegUnet); the few pairs the engine does not build (Linknet over EfficientNet,
ResNeSt under the dilated topologies) are refused loudly by ``vs_unet_create_ex`` rather than routed through a fallback."""
from __future__ import annotations

import logging
import os
from pathlib import Path
from typing import Tuple

import torch

from ..engine import VolSegUnet
from ..utilities import base_data_utils as utils


def _device(device_num) -> torch.device:
    if isinstance(device_num, torch.device):
        return device_num
    if isinstance(device_num, str):
        return torch.device(device_num)
    return torch.device("cuda", int(device_num))


def create_model_on_device(device_num: int, model_struc_dict: dict) -> torch.nn.Module:
    struct = dict(model_struc_dict)
    model_type = utils.create_enum_from_setting(struct.pop("type"), utils.ModelType)
    encoder = struct.get("encoder_name", "resnet34")
    topologies = {utils.ModelType.U_NET: "unet", utils.ModelType.U_NET_PLUS_PLUS: "unetplusplus", utils.ModelType.LINKNET: "linknet",
                  utils.ModelType.FPN: "fpn", utils.ModelType.DEEPLABV3_PLUS: "deeplabv3plus",
                  utils.ModelType.DEEPLABV3: "deeplabv3", utils.ModelType.MA_NET: "manet",
                  utils.ModelType.PAN: "pan"}
    if model_type not in topologies or encoder not in VolSegUnet.ENCODERS:
        raise NotImplementedError(
            f"the MI355X engine implements U_NET, U_NET_PLUS_PLUS, LINKNET, FPN, DEEPLABV3, DEEPLABV3_PLUS, MA_NET and PAN over {sorted(VolSegUnet.ENCODERS)} (requested "
            f"{model_type.name} + {encoder}); the other smp topologies / encoders are listed as next rows in SURVEY.md section 8f")
    if int(struct.get("in_channels", 1)) != 1:
        raise NotImplementedError("the engine implements the reference's single-channel input (config.MODEL_INPUT_CHANNELS)")
    model = VolSegUnet(int(struct["classes"]), device=_device(device_num), precision=struct.get("precision"), encoder=encoder,
                       topology=topologies[model_type])
    weights = struct.get("encoder_weights")
    if weights:
        # smp downloads ImageNet weights here; there is no network, so an explicit local torchvision state dict of the
        # same encoder can be supplied instead (env VOLSEG_ENCODER_WEIGHTS; VOLSEG_RESNET34_WEIGHTS for resnet34)
        path = os.environ.get("VOLSEG_ENCODER_WEIGHTS") or (os.environ.get("VOLSEG_RESNET34_WEIGHTS") if encoder == "resnet34" else None)
        if path and Path(path).exists():
            load_pretrained_encoder(model, torch.load(path, map_location="cpu"))
            logging.info(f"Loaded {weights} encoder weights from {path}")
        elif struct.get("allow_random_encoder"):
            logging.warning(f"encoder_weights={weights!r} requested but no local weights available "
                            "(set VOLSEG_ENCODER_WEIGHTS); allow_random_encoder is set: smp's random initialisation")
        else:
            # smp would download the ImageNet encoder here; the LR finder and the frozen-encoder phase are built around a
            # pretrained encoder, so silently training from random weights would be a behaviour change
            raise FileNotFoundError(
                f"encoder_weights={weights!r}: no network access and no local weights - point VOLSEG_ENCODER_WEIGHTS at a "
                f"torchvision {encoder} state dict, or set `allow_random_encoder: true` (or encoder_weights: null) in the "
                "model settings to train the encoder from random initialisation")
    logging.info(f"Sending the U-Net model to device {device_num}")
    return model


def load_pretrained_encoder(model: VolSegUnet, sd: dict) -> None:
    """Copy an encoder state dict (torchvision ResNet / ResNeXt, efficientnet-pytorch, timm ResNeSt - the packages smp
    takes its ImageNet weights from) into ``model.encoder``.  smp's ``patch_first_conv`` rule for ``in_channels=1``
    (segmentation_models_pytorch/encoders/_utils.py, called from model_2d.py:15-16 of the reference through
    ``smp.Unet(in_channels=1, encoder_weights="imagenet")``) is applied to whichever tensor IS the encoder's first
    convolution - ``conv1.weight`` (ResNets), ``_conv_stem.weight`` (EfficientNet), ``conv1.0.weight`` (ResNeSt's deep stem):
    a 3-channel kernel where this model holds a 1-channel one is summed over its input channels.  Keys may carry an
    ``encoder.`` prefix (an smp checkpoint) or not (the upstream package's own file); classifier heads are skipped."""
    own = model.state_dict()
    loaded = 0
    for k, v in sd.items():
        name = k if k.startswith("encoder.") else "encoder." + k
        if name not in own:
            continue  # fc.* / _fc.* / classifier heads
        tgt = own[name]
        if v.ndim == 4 and tgt.ndim == 4 and v.shape[1] == 3 and tgt.shape[1] == 1 and v.shape[0] == tgt.shape[0] and v.shape[2:] == tgt.shape[2:]:
            v = v.sum(1, keepdim=True)          # patch_first_conv: new_in_channels == 1
        if tuple(v.shape) != tuple(tgt.shape):
            raise ValueError(f"pretrained encoder tensor {k}: shape {tuple(v.shape)} does not fit {name} {tuple(tgt.shape)}")
        own[name] = v
        loaded += 1
    if not loaded:
        raise ValueError("the state dict holds no tensor of this encoder (wrong encoder_name for the weights file?)")
    model.load_state_dict(own)


load_torchvision_resnet34 = load_pretrained_encoder   # earlier name


def create_model_from_file(weights_fn: Path, gpu: bool = True, device_num: int = 0, precision: str | None = None) -> Tuple[torch.nn.Module, int, dict]:
    """Checkpoint -> (model, number of labels, label codes).  The checkpoint pickles the ModelType enum under the
    reference's module path, so ``weights_only=False`` and an importable ``volume_segmantics.utilities.base_data_utils``
    (or this package's alias, see checkpoint_compat) are needed for files written by the reference."""
    from ..checkpoint_compat import install_reference_aliases
    install_reference_aliases()
    weights_fn = Path(weights_fn).resolve()
    logging.info("Loading model dictionary from file.")
    model_dict = torch.load(weights_fn, map_location="cpu", weights_only=False)
    struct = dict(model_dict["model_struc_dict"])
    struct.update(model_dict.get("engine_settings", {}))   # keys only this engine knows (compute precision)
    struct["encoder_weights"] = None  # weights come from the file; never touch the network
    if precision:                      # the prediction settings' own `precision:` key (e.g. fp16 - inference only) wins over the
        struct["precision"] = precision    # precision the checkpoint was trained in
    model = create_model_on_device(device_num, struct)
    logging.info("Loading in the saved weights.")
    model.load_state_dict(model_dict["model_state_dict"])
    return model, model_dict["model_struc_dict"]["classes"], model_dict["label_codes"]
