"""Structural known answers for the restated smp U-Net/ResNet-34."""
import torch

from oracle.unet_resnet34_torch import OracleUnetResnet34, conv_macs_per_slice


def test_param_count_matches_published_smp_figure():
    # smp.Unet("resnet34") (3-channel input, 1 class) is published as 24,436,369 parameters
    assert sum(p.numel() for p in OracleUnetResnet34(3, 1).parameters()) == 24_436_369
    # 1-channel stem drops 2*64*49 weights
    assert sum(p.numel() for p in OracleUnetResnet34(1, 1).parameters()) == 24_436_369 - 2 * 64 * 49


def test_state_dict_keys_and_shapes():
    sd = OracleUnetResnet34(1, 4).state_dict()
    assert sd["encoder.conv1.weight"].shape == (64, 1, 7, 7)
    assert sd["encoder.layer2.0.downsample.0.weight"].shape == (128, 64, 1, 1)
    assert sd["encoder.layer4.2.conv2.weight"].shape == (512, 512, 3, 3)
    assert sd["decoder.blocks.0.conv1.0.weight"].shape == (256, 768, 3, 3)
    assert sd["decoder.blocks.3.conv1.0.weight"].shape == (32, 128, 3, 3)
    assert sd["decoder.blocks.4.conv1.0.weight"].shape == (16, 32, 3, 3)
    assert sd["decoder.blocks.4.conv2.1.num_batches_tracked"].shape == ()
    assert sd["segmentation_head.0.weight"].shape == (4, 16, 3, 3)
    assert sd["segmentation_head.0.bias"].shape == (4,)
    assert len(sd) == 278
    assert not any("downsample" in k for k in sd if k.startswith("encoder.layer1"))


def test_freeze_predicate_counts():
    # reference predicate: "encoder" in name and "conv" in name (vol_seg_2d_trainer.py:102-108;
    # KAT from tests/test_vol_seg_2d_trainer.py:37-44): stem + 2 convs per BasicBlock, not the 1x1 shortcuts
    names = [n for n, _ in OracleUnetResnet34(1, 2).named_parameters() if "encoder" in n and "conv" in n]
    assert len(names) == 1 + 2 * (3 + 4 + 6 + 3)
    assert not any("downsample" in n for n in names)


def test_macs_match_survey():
    assert abs(conv_macs_per_slice(256, 256, 2) / 1e9 - 7.721) < 1e-3
    assert abs(conv_macs_per_slice(512, 512, 4) / 1e9 - 30.958) < 1e-3


def test_forward_shape_multiple_of_32():
    net = OracleUnetResnet34(1, 3).eval()
    with torch.no_grad():
        assert net(torch.zeros(1, 1, 96, 64)).shape == (1, 3, 96, 64)


def test_other_resnet_encoders_match_torchvision_published_parameter_counts():
    """resnet18 / resnet34 / resnet50 behind the same U-Net decoder (oracle/unet_resnet_torch.py): the restated encoders carry
    exactly torchvision's published parameter counts once the 3-channel stem and the fc layer are accounted for, the tensor
    table of the engine's plan has the same keys / shapes in the same order, and resnet34 equals the original restatement."""
    from oracle.unet_resnet_torch import FC_PARAMS, TORCHVISION_PARAMS, OracleUnet
    from volume_segmantics_amd import _lib
    for name, enc in (("resnet18", 18), ("resnet34", 34), ("resnet50", 50), ("resnext50_32x4d", 51)):
        net = OracleUnet(name, 3, 2)
        n_enc = sum(p.numel() for p in net.encoder.parameters())
        assert n_enc + FC_PARAMS[name] == TORCHVISION_PARAMS[name], (name, n_enc)
        sd = OracleUnet(name, 1, 3).state_dict()
        table = _lib.unet_tensor_table(3, enc)
        keys = [k for k in sd if not k.endswith("num_batches_tracked")]
        assert [t[0] for t in table] == keys, name                                 # same names, same (state_dict) order
        assert all(tuple(sd[t[0]].shape) == tuple(t[1]) for t in table), name
    assert OracleUnet("resnet50", 1, 2).state_dict()["decoder.blocks.0.conv1.0.weight"].shape == (256, 2048 + 1024, 3, 3)
    assert OracleUnet("resnet50", 1, 2).state_dict()["encoder.layer1.0.downsample.0.weight"].shape == (256, 64, 1, 1)
    sd = OracleUnet("resnext50_32x4d", 1, 2).state_dict()     # groups = 32, width_per_group = 4: widths 128 .. 1024, 4 .. 32 per group
    assert sd["encoder.layer1.0.conv2.weight"].shape == (128, 4, 3, 3) and sd["encoder.layer4.2.conv2.weight"].shape == (1024, 32, 3, 3)
    a, b = OracleUnet("resnet34", 1, 2).state_dict(), OracleUnetResnet34(1, 2).state_dict()
    assert list(a) == list(b) and all(a[k].shape == b[k].shape for k in a)
    net = OracleUnet("resnet50", 1, 3).eval()
    with torch.no_grad():
        assert net(torch.zeros(1, 1, 64, 96)).shape == (1, 3, 64, 96)


def test_unetplusplus_matches_published_smp_parameter_counts():
    """smp.UnetPlusPlus (dense nested decoder, oracle/unet_resnet_torch.py:UnetPlusPlusDecoder): the restatement carries smp's
    published parameter counts - 26,078,609 for resnet34 and 48,985,745 for resnet50 (3-channel input, 1 class) - the engine's
    tensor table has the same keys / shapes in state_dict (registration) order, and a forward pass has the input's size."""
    from oracle.unet_resnet_torch import OracleUnet
    from volume_segmantics_amd import _lib
    assert sum(p.numel() for p in OracleUnet("resnet34", 3, 1, "unetplusplus").parameters()) == 26_078_609
    assert sum(p.numel() for p in OracleUnet("resnet50", 3, 1, "unetplusplus").parameters()) == 48_985_745
    for name, code in (("resnet18", 1018), ("resnet34", 1034), ("resnet50", 1050), ("resnext50_32x4d", 1051)):
        sd = OracleUnet(name, 1, 3, "unetplusplus").state_dict()
        table = _lib.unet_tensor_table(3, code)
        assert [t[0] for t in table] == [k for k in sd if not k.endswith("num_batches_tracked")], name
        assert all(tuple(sd[t[0]].shape) == tuple(t[1]) for t in table), name
    sd = OracleUnet("resnet34", 1, 2, "unetplusplus").state_dict()
    assert sd["decoder.blocks.x_0_0.conv1.0.weight"].shape == (256, 512 + 256, 3, 3)
    assert sd["decoder.blocks.x_1_3.conv1.0.weight"].shape == (64, 64 + 64 * 3, 3, 3)       # up(x_1_2) + [x_2_3, x_3_3, stem feature]
    assert sd["decoder.blocks.x_0_3.conv1.0.weight"].shape == (32, 64 + 64 * 4, 3, 3)
    assert sd["decoder.blocks.x_0_4.conv1.0.weight"].shape == (16, 32, 3, 3)
    keys = [k for k in sd if k.startswith("decoder.blocks.") and k.endswith("conv1.0.weight")]
    assert [k.split(".")[2] for k in keys] == ["x_0_0", "x_0_1", "x_1_1", "x_0_2", "x_1_2", "x_2_2", "x_0_3", "x_1_3", "x_2_3", "x_3_3", "x_0_4"]
    net = OracleUnet("resnet34", 1, 3, "unetplusplus").eval()
    with torch.no_grad():
        assert net(torch.zeros(1, 1, 96, 64)).shape == (1, 3, 96, 64)


def test_linknet_restatement_and_engine_table_agree():
    """smp.Linknet (oracle/unet_resnet_torch.py:LinknetDecoder): 1x1 convolution / ConvTranspose2d(4, 2, 1) / 1x1 convolution
    blocks with the encoder features added, 32 channels into a 1x1 head.  No published per-architecture parameter count is
    known here: the pins are the torchvision-pinned encoder, the decoder's arithmetic (487,232 parameters + a 33-parameter head
    on resnet34, 3-channel input, 1 class: 21,771,937 in all), the module key names of smp's nn.Sequential nesting, and that
    the engine's tensor table has the same keys / shapes in state_dict order for every encoder."""
    from oracle.unet_resnet_torch import OracleUnet
    from volume_segmantics_amd import _lib
    net = OracleUnet("resnet34", 3, 1, "linknet")
    assert sum(p.numel() for p in net.decoder.parameters()) == 487_232
    assert sum(p.numel() for p in net.parameters()) == 21_771_937
    for name, code in (("resnet18", 2018), ("resnet34", 2034), ("resnet50", 2050), ("resnext50_32x4d", 2051)):
        sd = OracleUnet(name, 1, 3, "linknet").state_dict()
        table = _lib.unet_tensor_table(3, code)
        assert [t[0] for t in table] == [k for k in sd if not k.endswith("num_batches_tracked")], name
        assert all(tuple(sd[t[0]].shape) == tuple(t[1]) for t in table), name
    sd = OracleUnet("resnet34", 1, 2, "linknet").state_dict()
    assert sd["decoder.blocks.0.block.0.0.weight"].shape == (128, 512, 1, 1)
    assert sd["decoder.blocks.0.block.1.0.weight"].shape == (128, 128, 4, 4) and sd["decoder.blocks.0.block.1.0.bias"].shape == (128,)
    assert sd["decoder.blocks.4.block.2.0.weight"].shape == (32, 16, 1, 1)
    assert sd["segmentation_head.0.weight"].shape == (2, 32, 1, 1)
    with torch.no_grad():
        assert net.eval()(torch.zeros(1, 3, 96, 64)).shape == (1, 1, 96, 64)


def test_fpn_restatement_and_engine_table_agree():
    """smp.FPN (oracle/unet_resnet_torch.py:FPNDecoder): 1,870,592 decoder parameters + a 129-parameter head on resnet34 (3-channel
    input, 1 class: 23,155,393 in all), smp's key nesting (decoder.p5, decoder.p{4,3,2}.skip_conv,
    decoder.seg_blocks.i.block.j.block.{0,1}), the engine's tensor table with the same keys / shapes in state_dict order for every
    encoder, logits at the input's size through the head's x4 bilinear upsampling."""
    from oracle.unet_resnet_torch import OracleUnet
    from volume_segmantics_amd import _lib
    net = OracleUnet("resnet34", 3, 1, "fpn")
    assert sum(p.numel() for p in net.decoder.parameters()) == 1_870_592
    assert sum(p.numel() for p in net.parameters()) == 23_155_393
    for name, code in (("resnet18", 3018), ("resnet34", 3034), ("resnet50", 3050), ("resnext50_32x4d", 3051)):
        sd = OracleUnet(name, 1, 3, "fpn").state_dict()
        table = _lib.unet_tensor_table(3, code)
        assert [t[0] for t in table] == [k for k in sd if not k.endswith("num_batches_tracked")], name
        assert all(tuple(sd[t[0]].shape) == tuple(t[1]) for t in table), name
    sd = OracleUnet("resnet50", 1, 2, "fpn").state_dict()
    assert sd["decoder.p5.weight"].shape == (256, 2048, 1, 1) and sd["decoder.p2.skip_conv.weight"].shape == (256, 256, 1, 1)
    assert sd["decoder.seg_blocks.0.block.2.block.0.weight"].shape == (128, 128, 3, 3)
    assert "decoder.seg_blocks.3.block.1.block.0.weight" not in sd and sd["decoder.seg_blocks.3.block.0.block.1.weight"].shape == (128,)
    assert sd["segmentation_head.0.weight"].shape == (2, 128, 1, 1)
    with torch.no_grad():
        assert net.eval()(torch.zeros(1, 3, 96, 64)).shape == (1, 1, 96, 64)


def test_deeplabv3plus_restatement_and_engine_table_agree():
    """smp.DeepLabV3Plus (oracle/unet_resnet_torch.py:DeepLabV3PlusDecoder, layer4 dilated by smp's replace_strides_with_dilation):
    1,152,528 decoder parameters + a 257-parameter head on resnet34 (3-channel input, 1 class: 22,437,457 in all - the 22.4 M of
    smp's model table), smp's key nesting (decoder.aspp.0.convs.{0..4}, .project, decoder.aspp.1 / .2, decoder.block1 / block2),
    the engine's tensor table with the same keys / shapes in state_dict order, dilation = padding = 2 on every 3x3 of layer4."""
    from oracle.unet_resnet_torch import OracleUnet
    from volume_segmantics_amd import _lib
    net = OracleUnet("resnet34", 3, 1, "deeplabv3plus")
    assert sum(p.numel() for p in net.decoder.parameters()) == 1_152_528
    assert sum(p.numel() for p in net.parameters()) == 22_437_457
    convs = [m for m in net.encoder.layer4.modules() if isinstance(m, torch.nn.Conv2d)]
    assert all(m.stride == (1, 1) and m.dilation == (2, 2) for m in convs)
    assert all(m.padding == ((2, 2) if m.kernel_size == (3, 3) else (0, 0)) for m in convs)
    for name, code in (("resnet18", 4018), ("resnet34", 4034), ("resnet50", 4050), ("resnext50_32x4d", 4051)):
        sd = OracleUnet(name, 1, 3, "deeplabv3plus").state_dict()
        table = _lib.unet_tensor_table(3, code)
        assert [t[0] for t in table] == [k for k in sd if not k.endswith("num_batches_tracked")], name
        assert all(tuple(sd[t[0]].shape) == tuple(t[1]) for t in table), name
    sd = OracleUnet("resnet34", 1, 2, "deeplabv3plus").state_dict()
    assert sd["decoder.aspp.0.convs.1.0.0.weight"].shape == (512, 1, 3, 3) and sd["decoder.aspp.0.convs.1.0.1.weight"].shape == (256, 512, 1, 1)
    assert sd["decoder.aspp.0.convs.4.1.weight"].shape == (256, 512, 1, 1) and sd["decoder.aspp.0.project.0.weight"].shape == (256, 1280, 1, 1)
    assert sd["decoder.block1.0.weight"].shape == (48, 64, 1, 1) and sd["decoder.block2.0.0.weight"].shape == (304, 1, 3, 3)
    assert sd["segmentation_head.0.weight"].shape == (2, 256, 1, 1)
    with torch.no_grad():
        assert net.eval()(torch.zeros(1, 3, 96, 64)).shape == (1, 1, 96, 64)


def test_deeplabv3_restatement_and_engine_table_agree():
    """smp.DeepLabV3 (oracle DeepLabV3Decoder; layer3 / layer4 dilated 2 / 4): 26,007,105 parameters on resnet34 (3-channel input,
    1 class) = the 26.0 M of smp's model table; decoder keys decoder.0.convs.*, decoder.0.project.*, decoder.1, decoder.2; the
    engine's tensor table agrees for every encoder it builds this topology for."""
    from oracle.unet_resnet_torch import OracleUnet
    from volume_segmantics_amd import _lib
    net = OracleUnet("resnet34", 3, 1, "deeplabv3")
    assert sum(p.numel() for p in net.parameters()) == 26_007_105
    for stage, rate in ((net.encoder.layer3, 2), (net.encoder.layer4, 4)):
        convs = [m for m in stage.modules() if isinstance(m, torch.nn.Conv2d)]
        assert all(m.stride == (1, 1) and m.dilation == (rate, rate) for m in convs)
    for name, code in (("resnet18", 5018), ("resnet34", 5034), ("resnet50", 5050), ("resnext50_32x4d", 5051)):
        sd = OracleUnet(name, 1, 3, "deeplabv3").state_dict()
        table = _lib.unet_tensor_table(3, code)
        assert [t[0] for t in table] == [k for k in sd if not k.endswith("num_batches_tracked")], name
        assert all(tuple(sd[t[0]].shape) == tuple(t[1]) for t in table), name
    sd = OracleUnet("resnet34", 1, 2, "deeplabv3").state_dict()
    assert sd["decoder.0.convs.1.0.weight"].shape == (256, 512, 3, 3) and sd["decoder.1.weight"].shape == (256, 256, 3, 3)
    with torch.no_grad():
        assert net.eval()(torch.zeros(1, 3, 96, 64)).shape == (1, 1, 96, 64)


def test_manet_restatement_and_engine_table_agree():
    """smp.MAnet (oracle MAnetDecoder: PAB centre + four MFAB blocks + a U-Net DecoderBlock): 31,783,633 parameters on resnet34
    (3-channel input, 1 class) = the 31.78 M of smp's model table; smp's key names (decoder.center.{top,center,bottom,out}_conv,
    decoder.blocks.i.{hl_conv,SE_ll,SE_hl,conv1,conv2}); the engine's tensor table agrees for every encoder."""
    from oracle.unet_resnet_torch import OracleUnet
    from volume_segmantics_amd import _lib
    net = OracleUnet("resnet34", 3, 1, "manet")
    assert sum(p.numel() for p in net.parameters()) == 31_783_633
    for name, code in (("resnet18", 6018), ("resnet34", 6034), ("resnet50", 6050), ("resnext50_32x4d", 6051)):
        sd = OracleUnet(name, 1, 3, "manet").state_dict()
        table = _lib.unet_tensor_table(3, code)
        assert [t[0] for t in table] == [k for k in sd if not k.endswith("num_batches_tracked")], name
        assert all(tuple(sd[t[0]].shape) == tuple(t[1]) for t in table), name
    sd = OracleUnet("resnet34", 1, 2, "manet").state_dict()
    assert sd["decoder.center.top_conv.weight"].shape == (64, 512, 1, 1) and sd["decoder.center.bottom_conv.bias"].shape == (512,)
    assert sd["decoder.blocks.0.hl_conv.1.0.weight"].shape == (256, 512, 1, 1) and sd["decoder.blocks.0.SE_ll.1.weight"].shape == (16, 256, 1, 1)
    assert sd["decoder.blocks.3.SE_hl.3.weight"].shape == (64, 4, 1, 1) and sd["decoder.blocks.0.conv1.0.weight"].shape == (256, 512, 3, 3)
    assert sd["decoder.blocks.4.conv1.0.weight"].shape == (16, 32, 3, 3)
    with torch.no_grad():
        assert net.eval()(torch.zeros(1, 3, 96, 64)).shape == (1, 1, 96, 64)


def test_pan_restatement_and_engine_table_agree():
    """smp.PAN (oracle PANDecoder; layer4 dilated 2): 21,475,816 parameters on resnet34 (3-channel input, 1 class) = the 21.5 M of
    smp's model table; keys decoder.fpa.{branch1.1,mid.0,down1.1,down2.1,down3.1,down3.2,conv2,conv1}.{conv,bn}, decoder.gau{3,2,1}.
    {conv1.1,conv2}.{conv,bn}; the engine's tensor table agrees; 7x7 / 5x5 / 3x3 single-channel kernels with biases."""
    from oracle.unet_resnet_torch import OracleUnet
    from volume_segmantics_amd import _lib
    net = OracleUnet("resnet34", 3, 1, "pan")
    assert sum(p.numel() for p in net.parameters()) == 21_475_816
    for name, code in (("resnet18", 7018), ("resnet34", 7034), ("resnet50", 7050), ("resnext50_32x4d", 7051)):
        sd = OracleUnet(name, 1, 3, "pan").state_dict()
        table = _lib.unet_tensor_table(3, code)
        assert [t[0] for t in table] == [k for k in sd if not k.endswith("num_batches_tracked")], name
        assert all(tuple(sd[t[0]].shape) == tuple(t[1]) for t in table), name
    sd = OracleUnet("resnet34", 1, 2, "pan").state_dict()
    assert sd["decoder.fpa.down1.1.conv.weight"].shape == (1, 512, 7, 7) and sd["decoder.fpa.conv2.conv.weight"].shape == (1, 1, 5, 5)
    assert sd["decoder.fpa.down3.2.bn.running_var"].shape == (1,) and sd["decoder.gau3.conv1.1.conv.bias"].shape == (32,)
    assert sd["decoder.gau1.conv2.conv.weight"].shape == (32, 64, 3, 3) and sd["segmentation_head.0.weight"].shape == (2, 32, 3, 3)
    with torch.no_grad():
        assert net.eval()(torch.zeros(1, 3, 128, 128)).shape == (1, 1, 128, 128)


def test_efficientnet_restatement_matches_published_parameter_counts_and_engine_table():
    """smp's efficientnet-b3 / b4 encoders (oracle/efficientnet_torch.py): with the 3-channel stem and the 1000-way `_fc` added back
    the restated encoders carry exactly efficientnet-pytorch's published parameter counts (12,233,232 / 19,341,616), their feature
    widths are smp's out_channels, the static same padding is (0, 1) / (1, 2) at stride 2, and the engine's tensor table has the
    same keys / shapes in state_dict order (incl. the never-run `_conv_head` / `_bn1`); the freeze predicate picks every
    convolution except the squeeze-excitation ones (their names lack "conv")."""
    from oracle.efficientnet_torch import COEFFS, OUT_CHANNELS, PUBLISHED_PARAMS, EfficientNetEncoder, SamePadConv2d, block_plan, round_filters
    from oracle.unet_resnet_torch import OracleUnet
    from volume_segmantics_amd import _lib
    for name, code in (("efficientnet-b3", 103), ("efficientnet-b4", 104)):
        enc = EfficientNetEncoder(name, 3)
        head = round_filters(1280, COEFFS[name][0])
        assert sum(p.numel() for p in enc.parameters()) + head * 1000 + 1000 == PUBLISHED_PARAMS[name], name
        with torch.no_grad():
            feats = enc.eval()(torch.zeros(1, 3, 64, 96))
        assert [f.shape[1] for f in feats[1:]] == list(OUT_CHANNELS[name][1:])
        assert [tuple(f.shape[2:]) for f in feats] == [(64 >> i, 96 >> i) for i in range(6)]
        for topology, tcode in (("unet", 0), ("unetplusplus", 1000), ("fpn", 3000), ("deeplabv3plus", 4000), ("deeplabv3", 5000), ("manet", 6000),
                                ("pan", 7000)):      # every decoder but Linknet (DeepLabV3+ / b4: BASELINE configs[4])
            sd = OracleUnet(name, 1, 3, topology).state_dict()
            table = _lib.unet_tensor_table(3, tcode + code)
            assert [t[0] for t in table] == [k for k in sd if not k.endswith("num_batches_tracked")], (name, topology)
            assert all(tuple(sd[t[0]].shape) == tuple(t[1]) for t in table), (name, topology)
        with torch.no_grad():      # output stride 16: the last stage keeps the 1/16 resolution (stride -> dilation 2)
            f = OracleUnet(name, 1, 2, "deeplabv3plus").eval().encoder(torch.zeros(1, 1, 64, 64))
        assert tuple(f[5].shape[2:]) == (4, 4) and tuple(f[4].shape[2:]) == (4, 4)
        frozen = [k for k, _ in OracleUnet(name, 1, 2).named_parameters() if "encoder" in k and "conv" in k]
        n_blocks, n_expand = len(block_plan(name)), sum(1 for b in block_plan(name) if b[2] != 1)
        assert len(frozen) == 2 + 2 * n_blocks + n_expand and not any("_se_" in k for k in frozen)
    for k, s, want in ((3, 2, (0, 1, 0, 1)), (5, 2, (1, 2, 1, 2)), (3, 1, (1, 1, 1, 1)), (5, 1, (2, 2, 2, 2)), (1, 1, (0, 0, 0, 0))):
        assert SamePadConv2d(8, 8, k, s, image_size=380).static_pad == want and SamePadConv2d(8, 8, k, s, image_size=300).static_pad == want


def test_resnest_restatement_matches_published_parameter_counts_and_engine_table():
    """smp's timm-resnest50d / timm-resnest101e encoders (oracle/resnest_torch.py, restated from timm 0.4.12): with the 3-channel stem and
    the 1000-way fc added back exactly timm's published parameter counts (27,483,240 / 48,275,016); feature widths = smp's out_channels;
    the engine's tensor table has the same keys / shapes in state_dict order under every decoder it builds them for (conv3 / bn3 in front
    of downsample.*, the two-group weight [2 C][C / 2][3][3]); the freeze predicate takes conv2's BatchNorms and biases too."""
    from oracle.resnest_torch import OUT_CHANNELS, PUBLISHED_PARAMS, ResNestEncoder
    from oracle.unet_resnet_torch import OracleUnet
    from volume_segmantics_amd import _lib
    for name, code in (("timm-resnest50d", 150), ("timm-resnest101e", 201)):
        enc = ResNestEncoder(name, 3)
        assert sum(p.numel() for p in enc.parameters()) + 2048 * 1000 + 1000 == PUBLISHED_PARAMS[name], name
        with torch.no_grad():
            feats = enc.eval()(torch.zeros(1, 3, 64, 96))
        assert [f.shape[1] for f in feats[1:]] == list(OUT_CHANNELS[name][1:])
        assert [tuple(f.shape[2:]) for f in feats] == [(64 >> i, 96 >> i) for i in range(6)]
        for topology, tcode in (("unet", 0), ("unetplusplus", 1000), ("linknet", 2000), ("fpn", 3000), ("manet", 6000)):
            sd = OracleUnet(name, 1, 3, topology).state_dict()
            table = _lib.unet_tensor_table(3, tcode + code)
            assert [t[0] for t in table] == [k for k in sd if not k.endswith("num_batches_tracked")], (name, topology)
            assert all(tuple(sd[t[0]].shape) == tuple(t[1]) for t in table), (name, topology)
        sd = OracleUnet(name, 1, 2).state_dict()
        assert sd["encoder.layer2.0.conv2.conv.weight"].shape == (256, 64, 3, 3) and sd["encoder.layer2.0.conv2.fc1.weight"].shape == (64, 128, 1, 1)
        assert sd["encoder.layer1.0.conv2.fc1.weight"].shape == (32, 64, 1, 1) and sd["encoder.layer1.0.downsample.1.weight"].shape[0] == 256
        frozen = [k for k, _ in OracleUnet(name, 1, 2).named_parameters() if "encoder" in k and "conv" in k]
        assert "encoder.layer1.0.conv2.bn0.weight" in frozen and "encoder.conv1.1.bias" in frozen and "encoder.layer1.0.conv2.fc2.bias" in frozen
        assert "encoder.layer1.0.bn1.weight" not in frozen and "encoder.layer1.0.downsample.1.weight" not in frozen and "encoder.bn1.weight" not in frozen
