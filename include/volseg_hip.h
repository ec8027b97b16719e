/* libvolseg_hip.so - C ABI of the MI355X-native (gfx950) 2D-slice segmentation engine.
 *
 * Drop-in boundary: the reference constructs the object behind ``self.model`` only in
 * ``create_model_on_device`` (volume_segmantics/model/model_2d.py:10-39) and uses it as
 * ``model(x)`` / ``loss.backward()`` (vol_seg_2d_trainer.py:424-430) and
 * ``model(batch)`` + softmax/argmax/gather/crop (vol_seg_2d_predictor.py:44-58), followed
 * by the max-probability merges (vol_seg_2d_predictor.py:90-98).  Every entry point below
 * cites the reference lines it replaces.  All pointers are DEVICE pointers unless named
 * ``host_*``; all activations are NHWC; nothing here allocates device memory or copies
 * host<->device; all work is ordered on the caller's HIP stream (``void* stream`` is a
 * hipStream_t; vs_unet_backward forks its weight-gradient kernels onto an internal side
 * stream and joins it back with events before returning, so stream order is preserved).  Functions return VS_OK (0) or a negative error code; the message is
 * available from vs_last_error() (thread local).
 */
#ifndef VOLSEG_HIP_H
#define VOLSEG_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { VS_OK = 0, VS_ERR_INVALID = -1, VS_ERR_HIP = -2, VS_ERR_UNSUPPORTED = -3 };
enum { VS_F32 = 0, VS_BF16 = 1, VS_F16 = 2 }; /* compute/storage dtype of activations and conv weights (VS_F16: inference only -
                                                  * the forward pass of a network in evaluation mode) */

const char* vs_last_error(void);
int vs_version(void);

/* ------------------------------------------------------------------------------------------
 * Single operators (unit-testable pieces of the network; torch equivalents in comments)
 * ---------------------------------------------------------------------------------------- */

/* Convolution geometry.  The input is the *virtual* tensor cat(upsample_nearest(src0, 2^up0),
 * src1) along channels - smp's DecoderBlock (F.interpolate(scale_factor=2, "nearest") +
 * torch.cat) is never materialised. */
typedef struct vs_conv_desc {
    int32_t dtype;             /* VS_F32 | VS_BF16 | VS_F16 */
    int32_t n, hin, win;       /* virtual input dims (after upsampling src0) */
    int32_t c0, c1, up0;       /* channels of src0 / src1 (0 = absent); src0: 0 = as is, 1 = nearest x2 upsampling,
                                  2 = zero stuffing x2 (values at the even rows / columns; c1 must be 0) */
    int32_t cout, kh, kw, stride, pad;
    int32_t relu;              /* apply ReLU in the epilogue */
    int32_t out_f32;           /* store fp32 output regardless of dtype */
    int32_t split_c;           /* >0: output channels >= split_c go to y1 (dgrad through a concat) */
    int32_t groups;            /* 0 / 1 = dense; > 1: grouped convolution (ResNeXt: nn.Conv2d(groups=32)) with c0 == cout and
                                  4 / 8 / 16 / 32 channels per group; w from vs_weights_prepare_grouped, dw [cout][taps][c0/groups] */
    int32_t dilation;          /* 0 / 1 = none; 2: a stride-1 3x3 convolution dilated by 2 (pad 2) - the stages torchvision / smp turn
                                  from stride into dilation (smp utils.replace_strides_with_dilation, DeepLabV3+ at output stride 16) */
} vs_conv_desc;

/* y = relu?( conv(x, w) * scale[c] + shift[c] + residual ).  w: [cout][kh*kw][c0+c1] in dtype.
 * Replaces F.conv2d (+ eval-mode F.batch_norm folded into scale/shift, + residual add, + ReLU)
 * inside smp.Unet.forward (reference call sites vol_seg_2d_trainer.py:424, vol_seg_2d_predictor.py:44).
 * Also used as dgrad (conv of dy with the flipped/transposed weights from vs_weights_prepare). */
int vs_conv2d_fwd(const vs_conv_desc* d, const void* src0, const void* src1, const void* w,
                  const float* scale, const float* shift, const void* residual,
                  void* y, void* y1, void* stream);
/* Which kernel instantiation vs_conv2d_fwd picks for a descriptor: cout tile * 1000 + pixel tiles per wave * 100 + taps * 10 + kind
 * (1 tile kernel, 2 stride 2, 4 direct shallow-layer kernel, 6 LDS-DMA ring, 7 persistent LDS-DMA ring, 8 eight-wave tiles).  No
 * reference counterpart: a test / tooling query (tests assert which path a parity case exercised). */
int vs_conv2d_variant(const vs_conv_desc* d);

/* Two chained evaluation-mode layers in ONE launch: y = act2(conv(act1(conv(up?(src0), w1) * scale1 + shift1), w2) * scale2 + shift2), both
 * 3x3 / stride 1 / padding 1, d1->cout == d2->c0 <= 16, d2->cout <= 16, 16-bit storage - smp's last decoder block at full resolution
 * (DecoderBlock.conv1 / conv2 with BatchNorm folded, inside model(batch) at vol_seg_2d_predictor.py:44).  The tensor between the two
 * layers is never written; every output bit equals two vs_conv2d_fwd launches.  vs_conv2d_pair_ok: whether the descriptors qualify
 * (1 / 0; the network plan asks the same question before it pairs two units). */
int vs_conv2d_pair_ok(const vs_conv_desc* d1, const vs_conv_desc* d2);
int vs_conv2d_pair_fwd(const vs_conv_desc* d1, const vs_conv_desc* d2, const void* src0, const void* w1, const float* scale1, const float* shift1,
                       const void* w2, const float* scale2, const float* shift2, void* y, void* stream);

/* The TRAINING forms of the same convolution launch - what the network plan (vs_unet_forward / vs_unet_backward) asks of it around a
 * train-mode BatchNorm2d; exposed as one operator so that every kernel the batch-32 step selects (LDS-DMA ring tiles of 64 / 32
 * couts, pairs of 8 x 8 images) can be held against torch with its training epilogue, at the launch sizes that select it.
 * Every group is optional (NULL / 0 = off):
 *  (a) statistics: the per-channel sum / sum of squares of the raw fp32 accumulators - F.batch_norm(training=True)'s batch
 *      statistics inside smp.Unet.forward (vol_seg_2d_trainer.py:424) - either as 64-bit fixed-point bins ([stats_nb][2][cout], sums
 *      scaled by vs_stat_scale(0) / (1), added atomically to what the caller zeroed; stats_nb a power of two) or as fp32 partial rows
 *      ([vs_conv2d_stat_rows][2][cout]);
 *  (b) pool0: the data gradient through nearest x2 upsampling (F.interpolate's backward, :429): channels < (y1 ? split_c : cout)
 *      are summed over 2 x 2 pixel blocks and stored to y at half resolution;
 *  (c) the first sweep of BatchNorm backward in the epilogue of the data gradient that completes a unit's activation gradient
 *      (autograd's batch-norm + ReLU backward, :429): g = (acc + residual) masked by (by > 0) - or, by == NULL, by the recomputed
 *      (bz - bmean) * binvstd * bgamma + bbeta > 0 - when brelu; y = g; bstats_partial [vs_conv2d_stat_rows][2][cout] = per-tile
 *      sum g, sum g * (bz - bmean) * binvstd;
 *  (d) normalise on load: src0 is the PRE-norm output of the producing unit whose statistics sit in nl_bins ([nl_nb][2][c0], nl_rows
 *      rows each); the launch finalises them (nl_mean / nl_invstd out, running statistics nl_rm / nl_rv updated when given),
 *      convolves relu((src0 - mean) * invstd * nl_gamma + nl_beta) rounded to the storage type, and leaves that activation in nl_y. */
typedef struct vs_conv_train {
    uint64_t* stats_bins; int32_t stats_nb; float* stats_partial;
    int32_t pool0;
    const void* bz; const void* by; const float* bmean; const float* binvstd; const float* bgamma; const float* bbeta;
    float* bstats_partial; int32_t brelu;
    const uint64_t* nl_bins; int32_t nl_nb; int64_t nl_rows; float nl_eps, nl_mom;
    float* nl_mean; float* nl_invstd; float* nl_rm; float* nl_rv; const float* nl_gamma; const float* nl_beta; void* nl_y;
} vs_conv_train;
int vs_conv2d_train(const vs_conv_desc* d, const void* src0, const void* src1, const void* w, const void* residual, void* y, void* y1,
                    const vs_conv_train* t, void* stream);
int vs_conv2d_train_variant(const vs_conv_desc* d, const vs_conv_train* t);   /* vs_conv2d_variant's code for that launch */
int vs_conv2d_stat_rows(const vs_conv_desc* d, const vs_conv_train* t);      /* partial rows that launch writes (one per tile) */
double vs_stat_scale(int which);   /* fixed-point scale of the bins: 0 = sum, 1 = sum of squares */

/* dw[cout][kh*kw][cin] (fp32) = sum_pixels dy (x) x.  torch: conv weight gradient of loss.backward()
 * (vol_seg_2d_trainer.py:429).  workspace >= vs_conv2d_wgrad_workspace(d). */
size_t vs_conv2d_wgrad_workspace(const vs_conv_desc* d);
/* The segmentation head's backward straight from dLoss / dlogits as autograd hands it over - fp32 NCHW planes of `classes` channels, no
 * 16-channel NHWC copy in between (replaces the head's share of loss.backward(), volume_segmantics/model/operations/vol_seg_2d_trainer.py:429).
 * vs_head_dgrad_planes: dx, the gradient of the head's 16-bit NHWC input [n][h][w][c] (c <= 16, a multiple of 4), from w = the forward's weight
 * copy [classes][9][c] in `dtype` (VS_BF16 / VS_F16); the (tap, class) pairs are the MFMA's K index.  VS_ERR_UNSUPPORTED unless 9 * classes <= 64.
 * vs_head_wgrad_planes: dw fp32 [classes][9][16] for a 16-channel bf16 input x; workspace from vs_head_wgrad_planes_workspace (0 = unsupported:
 * more than 7 classes, row widths other than 128 / 256 / 512 - the caller then converts and uses vs_conv2d_wgrad). */
int vs_head_dgrad_planes(int dtype, const float* dlogits, const void* w, void* dx, int n, int classes, int h, int wd, int c, void* stream);
size_t vs_head_wgrad_planes_workspace(int dtype, int n, int classes, int h, int wd);
int vs_head_wgrad_planes(int dtype, const void* x, const float* dlogits, float* dw, void* workspace, size_t workspace_bytes, int n, int classes,
                         int h, int wd, void* stream);
int vs_conv2d_wgrad(const vs_conv_desc* d, const void* src0, const void* src1, const void* dy,
                    float* dw, void* workspace, size_t workspace_bytes, void* stream);

/* fp32 master weights [cout][taps][cin] -> copy in `dtype` (wc, same layout; may be NULL) and the flipped / transposed copy
 * [cin][taps reversed][cout] the data gradient reads through vs_conv2d_fwd (wt; may be NULL).  The reference's autograd
 * derives both from the one weight tensor (loss.backward(), vol_seg_2d_trainer.py:429). */
int vs_weights_prepare(int dtype, const float* w, void* wc, void* wt, int cout, int taps, int cin, void* stream);
/* The same for a grouped convolution (cg = cin / groups channels per group, cin == cout): fp32 [cout][taps][cg] -> wc
 * [cout][taps][32] and wt [cin][taps reversed][32].  The kernels work on 32-channel super-groups; a narrower group is its
 * cg x cg block inside the super-group's 32 x 32 slab, zeros elsewhere (torchvision resnext50_32x4d: cg = 4, 8, 16, 32). */
int vs_weights_prepare_grouped(int dtype, const float* w, void* wc, void* wt, int cout, int taps, int cg, void* stream);

/* Stem: 7x7 stride-2 pad-3 conv with one input channel (encoder.conv1 after smp's
 * patch_first_conv).  x: [n][h][w] fp32 (the caller's (B,1,H,W) tensor); w: [64][49] fp32;
 * y: [n][h/2][w/2][64] in dtype; dy likewise. */
int vs_stem_fwd(int dtype, const float* x, const float* w, const float* scale, const float* shift, int relu,
                void* y, int n, int h, int w_, void* stream);
int vs_stem_wgrad(int dtype, const float* x, const void* dy, float* dw, float* workspace, size_t workspace_bytes,
                  int n, int h, int w_, void* stream);
size_t vs_stem_wgrad_workspace(int n, int h, int w_);

/* Train-mode BatchNorm2d (eps 1e-5, momentum 0.1, biased batch var, unbiased running var) on
 * x: [rows][c].  stats: mean[c], invstd[c]; running stats updated in place when not null. */
int vs_bn_stats(int dtype, const void* x, int64_t rows, int c, float eps, float momentum,
                float* mean, float* invstd, float* running_mean, float* running_var,
                float* workspace, size_t workspace_bytes, void* stream);
size_t vs_bn_workspace(int64_t rows, int c);
/* y = relu?( (x-mean)*invstd*gamma + beta + residual ) */
int vs_bn_apply(int dtype, const void* x, const float* mean, const float* invstd, const float* gamma,
                const float* beta, const void* residual, int relu, void* y, int64_t rows, int c, void* stream);
/* backward of the above: dz = dy * (y > 0 if relu); dgamma, dbeta; dx; optionally dres = dz. */
int vs_bn_bwd(int dtype, const void* dy, const void* y, const void* x, const float* mean, const float* invstd,
              const float* gamma, int relu, void* dx, void* dres, float* dgamma, float* dbeta,
              int64_t rows, int c, float* workspace, size_t workspace_bytes, void* stream);
/* same, but with y == NULL the ReLU mask is recomputed as (x-mean)*invstd*gamma + beta > 0 (units without a
 * residual input): two tensor reads fewer. */
int vs_bn_bwd_recompute(int dtype, const void* dy, const void* y, const void* x, const float* mean, const float* invstd,
                        const float* gamma, const float* beta, int relu, void* dx, void* dres, float* dgamma,
                        float* dbeta, int64_t rows, int c, float* workspace, size_t workspace_bytes, void* stream);
/* eval-mode folding: scale = gamma / sqrt(var + eps), shift = beta - mean * scale */
int vs_bn_fold(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
               float eps, float* scale, float* shift, int c, void* stream);

/* MaxPool2d(3, stride 2, pad 1) on NHWC; idx: uint8 [n][h/2][w/2][c] window position of the first max. */
int vs_maxpool_fwd(int dtype, const void* x, void* y, uint8_t* idx, int n, int h, int w, int c, void* stream);
int vs_maxpool_bwd(int dtype, const void* dy, const uint8_t* idx, void* dx, int accumulate,
                   int n, int h, int w, int c, void* stream);
/* backward of nearest x2 upsampling: dx[n][h][w][c] = sum of the 2x2 block of dy[n][2h][2w][c] */
int vs_upsample2x_bwd(int dtype, const void* dy, void* dx, int n, int h, int w, int c, void* stream);
/* dst[r][dst_off + k] (= | +=) src[r][src_off + k], k < c, for NHWC tensors viewed as [rows][c_src] / [rows][c_dst] (all channel
 * numbers multiples of 8).  torch.cat(..., dim=1) of smp's UnetPlusPlusDecoder (dense skips) in the forward pass, and the
 * accumulation of the concatenation's gradient slices onto its members in the backward pass. */
int vs_channel_slice(int dtype, const void* src, int c_src, int src_off, void* dst, int c_dst, int dst_off, int c,
                     int64_t rows, int accumulate, void* stream);
/* zero-stuffing for stride-2 dgrad: y[n][2h][2w][c] = x at even positions, 0 elsewhere */
int vs_zero_stuff2x(int dtype, const void* x, void* y, int n, int h, int w, int c, void* stream);

/* ------------------------------------------------------------------------------------------
 * Whole network: smp.Unet(resnet34, in_channels=1, classes=K)   (model_2d.py:15-16)
 * ---------------------------------------------------------------------------------------- */
typedef struct vs_unet vs_unet_t;

/* Parameter table (same for every instance with the same class count): tensors in smp's
 * state_dict order.  kind: 0 = conv weight (stored [cout][kh][kw][cin] = torch channels_last),
 * 1 = BN gamma, 2 = BN beta, 3 = conv bias, 4 = BN running_mean, 5 = BN running_var.
 * Kinds 0-3 live in the flat fp32 ``params`` buffer, kinds 4-5 in the flat ``bnstate`` buffer;
 * offset is in elements. */
int vs_unet_num_tensors(int classes);
int vs_unet_tensor_info(int classes, int index, char* name, int name_len, int64_t shape[4], int* ndim,
                        int* kind, int64_t* offset);
int64_t vs_unet_param_elems(int classes);
int64_t vs_unet_bnstate_elems(int classes);

int vs_unet_create(vs_unet_t** net, int dtype, int classes, int max_batch, int h, int w);
/* Other members of the reference's model matrix (model/model_2d.py:15-38, README.md:57-76).  `encoder` = topology * 1000 + depth:
 *   depth 18, 34 or 50: resnet18 / resnet34 (BasicBlock) / resnet50 (Bottleneck v1.5, expansion 4: 1x1 - 3x3(stride) - 1x1 plus a
 *   1x1 projection shortcut); 51: resnext50_32x4d (the same Bottleneck with groups = 32, width_per_group = 4: the 3x3
 *   convolution is grouped, 4 / 8 / 16 / 32 channels per group - see vs_weights_prepare_grouped);
 *   150 / 201: smp's timm-resnest50d / timm-resnest101e encoders (timm 0.4.12: deep stem, ResNestBottleneck with the radix-2 split-attention
 *   3x3 convolution, RadixSoftmax, avd / avg_down average pools) - topologies 0, 1, 2, 3, 6;
 *   103 / 104: smp's efficientnet-b3 / efficientnet-b4 encoders (efficientnet-pytorch 0.6.3: 3x3 / 2 stem, 26 / 32 MBConv blocks -
 *   expand 1x1, depthwise 3x3 / 5x5 behind static same padding, squeeze-excitation with swish, project 1x1, drop_connect + skip;
 *   BatchNorm2d(eps 1e-3, momentum 0.01) + swish; features (40, 32, 48, 136, 384) / (48, 32, 56, 160, 448)) - every topology but 2
 *   (under DeepLabV3+ / PAN / DeepLabV3 the last stage(s) run at stride 1 with dilation 2 / 2 / 2 and 4: smp's
 *   replace_strides_with_dilation);
 *   topology 7: smp.PAN (layer4 dilated; FPABlock with its single-channel 7x7 / 5x5 / 3x3 pyramid, three GAUBlocks, 3x3 head at 1/4
 *   resolution + x4 bilinear; slices must be multiples of 128) - depths 18 / 34 / 50;
 *   topology 6: smp.MAnet (PAB position attention at the deepest level, four MFAB blocks with squeeze-excitation gates on skip and
 *   input, a U-Net decoder block, 3x3 head);
 *   topology 5: smp.DeepLabV3 (output stride 8: layer3 / layer4 with dilation 2 / 4; dense dilated ASPP branches through
 *   vs_dilated_im2col; 3x3 conv; 1x1 head + x8 bilinear) - depths 18 / 34 / 50;
 *   topology 4: smp.DeepLabV3Plus (output stride 16: layer4 with dilation 2 instead of stride; ASPP with separable convolutions at
 *   rates 12 / 24 / 36 + image pooling, Dropout(0.5), x4 bilinear, 48-channel low-level branch, separable 3x3, 1x1 head + x4
 *   bilinear) - depths 18 / 34 / 50;
 *   topology 3: smp.FPN - biased 1x1 laterals + nearest-x2 top-down sums, Conv3x3 + GroupNorm(32) + ReLU + bilinear x2
 *   (align_corners) segmentation blocks, sum, Dropout2d(0.2), 1x1 head at 1/4 resolution + bilinear x4;
 *   topology 2: smp.Linknet - 1x1 convolution / ConvTranspose2d(4, 2, 1) / 1x1 convolution blocks, encoder features added;
 *   topology 0: smp.Unet; 1: smp.UnetPlusPlus - the dense nested decoder (node x_d_l = DecoderBlock(up(x_d_(l-1)),
 *   cat(x_(d+1)_l .. x_l_l, encoder feature)); the concatenations are materialised by vs_channel_slice copies and their
 *   gradients accumulated back onto the members).
 * The plain entry points above are encoder = 34 (U-Net / resnet34). */
int vs_unet_create_ex(vs_unet_t** net, int dtype, int classes, int max_batch, int h, int w, int encoder);
int vs_unet_num_tensors_ex(int classes, int encoder);
int vs_unet_tensor_info_ex(int classes, int encoder, int index, char* name, int name_len, int64_t shape[4], int* ndim,
                           int* kind, int64_t* offset);
int64_t vs_unet_param_elems_ex(int classes, int encoder);
int64_t vs_unet_bnstate_elems_ex(int classes, int encoder);
void vs_unet_destroy(vs_unet_t* net);
size_t vs_unet_workspace_bytes(const vs_unet_t* net, int training);
/* (re)derive low-precision / transposed weight copies and folded BN constants from the fp32
 * master parameters; call after every optimizer step or load_state_dict. */
int vs_unet_prepare(vs_unet_t* net, const float* params, const float* bnstate, int training, void* workspace,
                    void* stream);
/* logits (n, K, h, w) fp32 NCHW = model(x), x: (n, 1, h, w) fp32.  training != 0: batch-stat BN,
 * running stats in ``bnstate`` updated, activations kept in ``workspace`` for vs_unet_backward.
 * Replaces self.model(inputs) (vol_seg_2d_trainer.py:232,424; vol_seg_2d_predictor.py:44). */
int vs_unet_forward(vs_unet_t* net, const float* params, float* bnstate, const float* x, int n, int training,
                    float* logits, void* workspace, void* stream);
/* grads (flat fp32, same layout as params, overwritten) = d loss / d params given dlogits
 * (n, K, h, w) fp32 NCHW.  need_encoder_wgrad = 0 skips the weight gradients of the tensors
 * the reference freezes (vol_seg_2d_trainer.py:102-108).  Replaces loss.backward() (:429). */
int vs_unet_backward(vs_unet_t* net, const float* params, const float* x, const float* dlogits, int n,
                     int need_encoder_wgrad, float* grads, void* workspace, void* stream);

/* The same backward restricted to the plan's units [unit_lo, unit_hi), for callers that overlap a bucketed gradient
 * all-reduce with the rest of the backward pass: call it for consecutive ranges from the top (unit_hi = vs_unet_num_units)
 * down to 0.  On return the gradients of the range are ordered on `stream`.  vs_unet_unit_param_offset(net, u) is the
 * element offset in the flat parameter / gradient buffer where unit u's tensors start (u = num_units: the total). */
int vs_unet_backward_range(vs_unet_t* net, const float* params, const float* x, const float* dlogits, int n,
                           int need_encoder_wgrad, float* grads, void* workspace, void* stream, int unit_lo, int unit_hi);
int64_t vs_unet_unit_param_offset(const vs_unet_t* net, int unit);

/* Diagnostics: where a unit's activation (a), pre-BN output (z) and their gradients (da, dz) live
 * inside the workspace (byte offsets; NHWC, dtype of the plan).  Tests only. */
int vs_unet_num_units(const vs_unet_t* net);
/* Normalise-on-load plan of a bf16 TRAINING forward at batch n: flags[i] = 1 when unit i (a convolution + BatchNorm + ReLU whose
 * output has exactly one reader, a stride-1 3x3 convolution) will NOT run a normalisation sweep - the reader sums the unit's statistics
 * bins itself, normalises the pre-norm tensor while staging it and stores the activation as a by-product.  Replaces, for
 * those units, torch's BatchNorm2d + ReLU modules between two convolutions (reference call site vol_seg_2d_trainer.py:424, the
 * model's forward).  Host logic only; returns the number of units. */
int vs_unet_nl_plan(vs_unet_t* net, int n, int* flags, int cap);
/* SyncBatchNorm for data-parallel training (N ranks reproduce the reference's single BatchNorm batch - data/dataloaders.py:42-49, one
 * loader feeding one model): `hook(user, values, count, kind, stream)` must sum `count` device values in place over the ranks,
 * ordered on `stream` (kind 0: int64 - the fixed-point statistics sums, exact; kind 1: fp32), and return 0.  `world` ranks
 * contribute equal shares of the global batch.  hook = NULL restores per-rank statistics.  bf16 plans whose BatchNorms all sit
 * behind bias-free convolutions / the ResNet stem; others are refused. */
typedef int (*vs_stats_hook_t)(void* user, void* values, int64_t count, int kind, void* stream);
int vs_unet_set_stats_hook(vs_unet_t* net, vs_stats_hook_t hook, void* user, int world);
int vs_unet_debug_unit(const vs_unet_t* net, int unit, char* weight_name, int name_len, int* c, int* h, int* w,
                       size_t* off_a, size_t* off_z, size_t* off_da, size_t* off_dz);

/* Per-kernel-class timing with HIP events recorded on the launch stream (bench.py's roofline block).
 * vs_profile_enable(1) clears and starts collecting; vs_profile_read sums, per kind, the elapsed ms, the
 * algorithmic flops / bytes and the number of launches recorded since then (arrays of vs_profile_num_kinds()). */
int vs_profile_enable(int on);
int vs_profile_enabled(void);
int vs_profile_num_kinds(void);
const char* vs_profile_kind_name(int kind);
int vs_profile_read(double* ms, double* flops, double* bytes, int64_t* calls);
/* raw records in launch order (tag = unit index inside the network plan); returns the count or -1 */
int vs_profile_read_raw(int max_n, int* kind, int* tag, int* variant, double* ms, double* flops, double* bytes);
/* Diagnostics only (tools/conv_probe.py): while `buf` (device memory, 8 x u64 per workgroup, `cap_wgs` workgroups)
 * is set, convolution launches of at most cap_wgs workgroups record per-workgroup phase timestamps.  NULL disables. */
int vs_debug_probe(void* buf, size_t cap_wgs);

/* Runtime options (csrc/prof.hip holds the table): eleven kernel-family switches - "side_stream", "conv_direct", "conv_nw8", "conv_ring",
 * "conv_stream", "wgrad_ring", "wgrad_xcd", "stats_bins", "fuse_bn_bwd", "nl_fwd", "stem_bf16" (all 1) - and thirteen launch-size
 * thresholds / split sizes ("wgrad_target" 96, "conv_min_wgs" 512, "fork_every" 2, ...); initial values can come from the
 * environment as VS_<NAME>.  No reference counterpart: tests and tooling. */
int vs_set_option(const char* name, int value);
int vs_get_option(const char* name);

/* AdamW over a flat fp32 buffer (torch.optim.AdamW semantics, vol_seg_2d_trainer.py:395-396,430);
 * ``mask`` (uint8 per element, may be null) = 0 freezes an element (requires_grad False). */
int vs_adamw_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, const uint8_t* mask,
                  int64_t n, float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                  void* stream);

/* The same update folded into the backward pass: every convolution's AdamW step (and the derivation of its low-precision
 * / transposed copies for the next forward, into the plan's second weight set) is queued behind that layer's weight
 * gradient on the library's side stream, i.e. in the shadow of the remaining backward work; the small tensors (BN affine
 * parameters, head bias) are updated in one launch at the end.  Numerically identical to vs_unet_backward followed by
 * vs_adamw_step with a mask that skips the frozen encoder convolutions (need_encoder_wgrad == 0).  After it returns (all
 * work ordered on `stream`) the caller must NOT call vs_unet_prepare for this plan again before the next forward. */
typedef struct vs_adamw_args {
    float* params;                 /* flat fp32 parameter buffer (updated in place; also the `params` the forward read) */
    float* exp_avg;
    float* exp_avg_sq;
    float lr, beta1, beta2, eps, weight_decay;
    int32_t step;                  /* 1-based number of this update (bias correction) */
    const float* hyper;            /* optional, DEVICE: 8 floats written by vs_train_hyper_set; when set the kernels read lr, beta1,
                                      beta2, eps, weight_decay and the bias corrections from there instead of the fields above, so a
                                      captured step (vs_capture_begin) can be replayed under a scheduler (OneCycleLR moves lr and
                                      beta1 every step, vol_seg_2d_trainer.py:401-408) */
} vs_adamw_args;
int vs_unet_backward_adamw(vs_unet_t* net, const float* x, const float* dlogits, int n, int need_encoder_wgrad,
                           float* grads, void* workspace, void* stream, const vs_adamw_args* opt);

/* Data-parallel form of the same idea, driven by the caller bucket by bucket: after a bucket of gradients has been
 * all-reduced and vs_adamw_step has updated that slice of the parameters, vs_unet_prepare_range derives the weight copies of
 * the bucket's units [unit_lo, unit_hi) into the plan's second weight set; vs_unet_flip_weight_set makes that set the one the
 * next forward reads (call it once, after every range of the network has been refreshed). */
int vs_unet_prepare_range(vs_unet_t* net, const float* params, void* workspace, void* stream, int unit_lo, int unit_hi);
int vs_unet_flip_weight_set(vs_unet_t* net);
/* The weight-gradient work of vs_unet_backward* runs on library-owned side streams (one pair per device for every plan of the
 * process).  HIP multiplexes streams onto a few hardware queues; two streams on one queue run in order.  The library therefore
 * checks, the first time the streams serve a caller's stream, that a kernel on each really runs beside the other (two 100 us spin
 * kernels, timed) and replaces a stream that does not.  This entry repeats that measurement for `stream`: *overlaps = 1 when the
 * plan's weight-gradient stream and `stream` executed concurrently.  (No reference counterpart: torch's autograd engine runs on
 * the caller's stream only.) */
int vs_unet_side_stream_overlaps(vs_unet_t* net, void* stream, int* overlaps);
/* The library's side stream `index` (0 / 1) itself, verified against `stream` as above - for a host that replays recorded shares of the
 * step (vs_unet_backward_adamw_part, role 2) or runs its own side work and wants a stream known to overlap the caller's. */
int vs_unet_side_stream(vs_unet_t* net, void* stream, int index, void** side);
int vs_unet_weight_set(const vs_unet_t* net);   /* 0 / 1: the weight set the next forward reads */

/* ------------------------------------------------------------------------------------------
 * Captured steps: the host loop of _train_one_batch (vol_seg_2d_trainer.py:419-432) as a replayed hipGraph
 * ---------------------------------------------------------------------------------------- */
/* vs_capture_begin returns the library's capture stream (NULL on error): every library call made with it as `stream`
 * until vs_capture_end is RECORDED, not executed - including the backward pass's side-stream branches.  vs_graph_launch
 * replays the recorded sequence, ordered on `stream`.  The recorded calls keep their pointer arguments: the buffers must
 * stay in place and hold the new inputs before each replay.  A training step recorded with vs_unet_backward_adamw flips
 * the plan's weight set: record one graph per vs_unet_weight_set value, and call vs_unet_flip_weight_set after each replay
 * (a replay runs no host code of the library). */
typedef struct vs_graph vs_graph_t;
void* vs_capture_begin(void);
int vs_capture_end(vs_graph_t** out);
int vs_capture_abort(void);
int vs_graph_launch(vs_graph_t* g, void* stream);
int64_t vs_graph_num_nodes(const vs_graph_t* g);
void vs_graph_destroy(vs_graph_t* g);
/* One stream's share of vs_unet_backward_adamw for the units [unit_lo, unit_hi) (top of the network first, as
 * vs_unet_backward_range), enqueued in order on `stream`: role 1 = the caller's-stream kernels (BatchNorm backward, data
 * gradients), role 2 = weight gradients, AdamW and the next forward's weight copies.  Recorded separately, each share is a
 * LINEAR graph, which the runtime replays as one batch of queue packets (0.13 ms of host time per step; a graph with
 * parallel branches is walked node by node, 1-2.5 ms).  There are no events inside: the caller runs role 2 of a range behind
 * role 1 of the same range (an event between two streams), range by range.  Neither role flips the weight set. */
int vs_unet_backward_adamw_part(vs_unet_t* net, const float* x, const float* dlogits, int n, int need_encoder_wgrad,
                                float* grads, void* workspace, void* stream, const vs_adamw_args* opt, int unit_lo,
                                int unit_hi, int role);
/* Pixel shuffle with block 2 and its inverse (NHWC): y[n][2i+a][2j+b][k] = x[n][i][j][(2a+b) * c + k] (+ bias[k]) (* scale[k] +
 * shift[k]) (ReLU), each part optional (NULL / 0).  nn.ConvTranspose2d(kernel 4, stride 2, padding 1) - smp Linknet's
 * TransposeX2 (decoders/linknet/decoder.py) - is a 3x3 convolution onto 4 * c channels (vs_conv2d_fwd; parity (a, b) of the
 * output reads taps kh = a + 3 - 2r, kw = b + 3 - 2q of the 4x4 kernel) followed by this shuffle.  vs_colsum: out[k] = sum over
 * rows of x[rows][c] in fp32, fixed order (the bias gradient); workspace vs_colsum_workspace(c) bytes. */
int vs_depth_to_space2(int dtype, const void* x, void* y, int n, int h, int w, int c, const float* bias, const float* scale,
                       const float* shift, int relu, void* stream);
int vs_space_to_depth2(int dtype, const void* x, void* y, int n, int h, int w, int c, void* stream);
/* the transposed convolution's weights in the 3x3 form: w fp32 [cin][cout][4][4] (torch's layout) -> wc [4 * cout][9][cin]
 * (may be NULL) and the data-gradient copy wt [cin][9 reversed][4 * cout] (may be NULL); vs_convt_wgrad_gather maps the 3x3
 * form's weight gradient (vs_conv2d_wgrad: [4 * cout][9][cin] fp32) back to dw [cin][cout][4][4] */
int vs_convt_weights_prepare(int dtype, const float* w, void* wc, void* wt, int cin, int cout, void* stream);
int vs_convt_wgrad_gather(const float* dense, float* dw, int cin, int cout, void* stream);
size_t vs_colsum_workspace(int c);
int vs_colsum(int dtype, const void* x, int64_t rows, int c, float* out, float* workspace, size_t workspace_bytes, void* stream);
/* ---- streaming operators of smp.FPN's decoder (segmentation-models-pytorch 0.2.1, decoders/fpn/decoder.py; model/model_2d.py:22 of the
 * reference builds smp.FPN(**model_struc_dict)); NHWC tensors in `dtype`, c a multiple of 8 ------------------------------------
 * vs_upsample2x_add: y = F.interpolate(x, scale_factor=2, mode="nearest") + skip (FPNBlock); x [n][h][w][c], skip / y [n][2h][2w][c].
 * vs_gn_fwd / vs_gn_bwd: nn.GroupNorm(groups, c) (+ ReLU) of Conv3x3GNReLU on x [n][hw][c]: stats [n][groups][2] = {mean,
 *   1/sqrt(var + eps)} fp32 (written by fwd, read by bwd); bwd takes the gradient w.r.t. the (post-ReLU) output and masks it
 *   itself (recomputed from x), writes dx, dgamma[c], dbeta[c].  c <= 256 with 256 % (c / 8) == 0.
 * vs_bilinear_up(_bwd): F.interpolate(scale_factor=factor, mode="bilinear", align_corners=True) and its adjoint (accumulate = 1
 *   adds to dx); vs_bilinear_up_planes(_bwd): the same on fp32 planes [planes][h][w] - nn.UpsamplingBilinear2d of the
 *   SegmentationHead on NCHW logits.
 * vs_dropout2d_mask + vs_channel_scale: nn.Dropout2d(p): mask [n][c] = 0 or 1/(1-p), drawn per (sample, channel) from a
 *   counter-based generator keyed by (seed, *counter + bias) - counter is a device int64 (may be NULL) so that a replayed graph
 *   draws a fresh mask per step; vs_channel_scale multiplies x [n][hw][c] by it (the forward, and the backward on the gradient). */
int vs_upsample2x_add(int dtype, const void* x, const void* skip, void* y, int n, int h, int w, int c, void* stream);
size_t vs_gn_workspace(int n, int c);
size_t vs_gn_bwd_workspace(int n, int c, int groups);
int vs_gn_fwd(int dtype, const void* x, const float* gamma, const float* beta, int relu, void* y, float* stats, int n, int64_t hw,
              int c, int groups, float eps, float* workspace, size_t workspace_bytes, void* stream);
int vs_gn_bwd(int dtype, const void* dy, const void* x, const float* stats, const float* gamma, const float* beta, int relu, void* dx,
              float* dgamma, float* dbeta, int n, int64_t hw, int c, int groups, float* workspace, size_t workspace_bytes, void* stream);
int vs_bilinear_up(int dtype, const void* x, void* y, int n, int h, int w, int c, int factor, void* stream);
int vs_bilinear_up_bwd(int dtype, const void* dy, void* dx, int n, int h, int w, int c, int factor, int accumulate, void* stream);
int vs_bilinear_up_planes(const float* x, float* y, int planes, int h, int w, int factor, void* stream);
int vs_bilinear_up_planes_bwd(const float* dy, float* dx, int planes, int h, int w, int factor, void* stream);
int vs_dropout2d_mask(float* mask, int n, int c, float p, uint32_t seed, const int64_t* counter, int64_t bias, void* stream);
int vs_channel_scale(int dtype, const void* x, const float* mask, void* y, int n, int64_t hw, int c, void* stream);

/* ---- the non-dense operators of smp's DeepLabV3+ decoder (decoders/deeplabv3/decoder.py), NHWC, c a multiple of 8 --------------------
 * vs_dwconv3x3: nn.Conv2d(c, c, 3, padding=d, dilation=d, groups=c, bias=False) - the depthwise half of SeparableConv2d; w fp32
 *   [c][9]; flip = 1 gives the data gradient.  vs_dwconv3x3_wgrad: dw [c][9] (fp32).
 * vs_spatial_sum / vs_broadcast_rows: nn.AdaptiveAvgPool2d(1) (scale 1 / hw) and F.interpolate of the 1x1 map back to the
 *   feature's size in ASPPPooling - and each other's gradients.
 * vs_dropout: element-wise nn.Dropout(p) of ASPP.project, mask = f(seed, *counter + bias, element) recomputed per call. */
int vs_dwconv3x3(int dtype, const void* x, const float* w, void* y, int n, int h, int wd, int c, int dilation, int flip, void* stream);
size_t vs_dwconv3x3_wgrad_workspace(int c);
int vs_dwconv3x3_wgrad(int dtype, const void* x, const void* dy, float* dw, int n, int h, int wd, int c, int dilation, float* workspace,
                       size_t workspace_bytes, void* stream);
int vs_spatial_sum(int dtype, const void* x, void* y, int n, int64_t hw, int c, float scale, void* stream);
int vs_broadcast_rows(int dtype, const void* v, void* y, int n, int64_t hw, int c, float scale, int accumulate, void* stream);
int vs_dropout(int dtype, const void* x, void* y, int64_t elems, float p, uint32_t seed, const int64_t* counter, int64_t bias, void* stream);
/* a 3x3 convolution at a large dilation r (padding r) as ONE 1x1 convolution over 9 c channels: col[n][i][j][tap][c] =
 * x[n][i + (kh - 1) r][j + (kw - 1) r][c] (zero beyond the map; the weights [cout][tap][c] are that 1x1 convolution's matrix as they
 * lie); inverse = 1 is the adjoint, dst[n][i][j][c] (+)= the sum over taps of src's shifted entries (DeepLabV3's dense ASPP branches
 * at rates 12 / 24 / 36, where most taps of a 32 x 32 map look at padding) */
int vs_dilated_im2col(int dtype, const void* src, void* dst, int n, int h, int w, int c, int r, int inverse, int accumulate, void* stream);

/* ---- what smp's EfficientNet encoders add (encoders/efficientnet.py over efficientnet-pytorch 0.6.3: model.py MBConvBlock, utils.py
 * Conv2dStaticSamePadding / drop_connect), NHWC (csrc/effnet.hip) -------------------------------------------------------------------------
 * vs_bn2_*: nn.BatchNorm2d(c, eps, momentum) for ANY c that is a multiple of 8 (the expanded widths reach 2688), followed by nothing
 *   (act 0), ReLU (1) or swish x * sigmoid(x) (2); bwd recomputes the activation's derivative from x.  vs_bn2_apply with var_eps >= 0
 *   reads VARIANCES in place of invstd (evaluation straight from the running statistics).  workspace: vs_bn2_workspace(c) bytes.
 * vs_dwconv2d*: nn.Conv2d(c, c, k, stride, groups=c, bias=False), k 3 / 5, stride 1 / 2, pad_lo zero rows / columns in front (TF "same"
 *   static padding: (0, 1) for k = 3 and (1, 2) for k = 5 at stride 2; dilation > 1 with pad_lo = (k / 2) * dilation: what smp's
 *   replace_strides_with_dilation leaves of a stage under DeepLabV3+); w fp32 [c][k * k].  x_single_channel = 1: x is an fp32 [n][h][w]
 *   map broadcast over the channels = nn.Conv2d(1, c, k, stride, bias=False), the stem on greyscale slices.
 * vs_sample_scale_add: y = x * mask[n] + skip - drop_connect (mask from vs_dropout2d_mask with c = 1) and the residual sum in one sweep;
 *   on the gradient (skip NULL) its backward.  vs_sample_rowsum: out[n][c] = scale * sum over hw of a (* b) for any c % 8 == 0. */
size_t vs_bn2_workspace(int c);
int vs_bn2_stats(int dtype, const void* x, int64_t rows, int c, float eps, float momentum, float* mean, float* invstd, float* running_mean,
                 float* running_var, float* workspace, size_t workspace_bytes, void* stream);
int vs_bn2_apply(int dtype, const void* x, const float* mean, const float* invstd, const float* gamma, const float* beta, int act, float var_eps,
                 void* y, int64_t rows, int c, void* stream);
int vs_bn2_bwd(int dtype, const void* dy, const void* x, const float* mean, const float* invstd, const float* gamma, const float* beta, int act,
               void* dx, float* dgamma, float* dbeta, int64_t rows, int c, float* workspace, size_t workspace_bytes, void* stream);
int vs_dwconv2d(int dtype, const void* x, const float* w, void* y, int n, int h, int wd, int c, int k, int stride, int pad_lo, int dilation, int ho,
                int wo, int x_single_channel, void* stream);
int vs_dwconv2d_bwd_data(int dtype, const void* dy, const float* w, void* dx, int n, int h, int wd, int c, int k, int stride, int pad_lo, int dilation,
                         int ho, int wo, int accumulate, void* stream);
/* evaluation: depthwise convolution + folded BatchNorm (scale / shift from vs_bn_fold) + activation (0 / 1 ReLU / 2 swish) in one sweep */
int vs_dwconv2d_affine(int dtype, const void* x, const float* w, const float* scale, const float* shift, int act, void* y, int n, int h, int wd, int c,
                       int k, int stride, int pad_lo, int dilation, int ho, int wo, void* stream);
size_t vs_dwconv2d_wgrad_workspace(int c, int k);
int vs_dwconv2d_wgrad(int dtype, const void* x, const void* dy, float* dw, int n, int h, int wd, int c, int k, int stride, int pad_lo, int dilation, int ho,
                      int wo, int x_single_channel, float* workspace, size_t workspace_bytes, void* stream);
int vs_sample_scale_add(int dtype, const void* x, const float* mask, const void* skip, void* y, int n, int64_t per_sample, void* stream);
int vs_sample_rowsum(int dtype, const void* a, const void* b, void* out, int n, int64_t hw, int c, float scale, void* stream);
/* the same with a sample's rows spread over up to 64 workgroups and a fixed-order finish (few samples, large maps: prediction batches) */
size_t vs_sample_rowsum_workspace(int n, int c);
int vs_sample_rowsum_ws(int dtype, const void* a, const void* b, void* out, int n, int64_t hw, int c, float scale, float* workspace,
                        size_t workspace_bytes, void* stream);

/* ---- timm's ResNeSt blocks (timm-resnest50d / timm-resnest101e behind smp; csrc/resnest.hip): RadixSoftmax(radix 2, cardinality 1) on
 * attention logits z [n][2 c] (the two splits' values c apart): a = softmax over each pair; bwd: dz from da and a.  The rest of a block is
 * composed from the operators above (the two-group 3x3 convolution as a dense one with block-expanded weight copies). */
int vs_radix2_softmax(int dtype, const void* z, void* a, int n, int c, void* stream);
int vs_radix2_softmax_bwd(int dtype, const void* da, const void* a, void* dz, int n, int c, void* stream);
/* SplitAttnConv2d's last step in one sweep: out [n][hw][c] = x[.., ch] a[n][ch] + x[.., c + ch] a[n][c + ch] (x [n][hw][2 c], a [n][2 c]); bwd: dx from
 * dout.  vs_sample_rowsum_b: out [n][c] = sum over hw of a [n][hw][c] * b [n][hw][cb], b repeated across the c / cb blocks (the attention's gradient). */
int vs_radix2_gated_sum(int dtype, const void* x, const void* a, void* out, int n, int64_t hw, int c, void* stream);
int vs_radix2_gated_sum_bwd(int dtype, const void* dout, const void* a, void* dx, int n, int64_t hw, int c, void* stream);
int vs_sample_rowsum_b(int dtype, const void* a, const void* b, int cb, void* out, int n, int64_t hw, int c, float* workspace, size_t workspace_bytes,
                       void* stream);

/* ---- the attention operators of smp.MAnet's decoder (decoders/manet/decoder.py), NHWC ---------------------------------------------------
 * vs_pab_attention_fwd/bwd: PAB - sp = softmax over ALL hw x hw entries of center top^T (top, center [n][hw][K]), out = sp bottom
 *   ([n][hw][C]), y = x + the product's memory reinterpreted as (n, C, h, w) exactly as smp's reshape does; sp fp32 [n][hw][hw] is
 *   kept for bwd, which returns the gradients of the attention term w.r.t. top, center, bottom (the identity path is the caller's).
 *   scratch: vs_pab_scratch_bytes(n, hw, C).
 * vs_se_gate_fwd/bwd: MFAB's SE_ll / SE_hl after the average pool: a = sigmoid(W2 relu(W1 p + b1) + b2) on p [n][C]; W1 [R][C], W2
 *   [C][R], b1, b2 fp32 (torch's 1x1 Conv2d weights); hid [n][R] fp32 carries the hidden layer to bwd (dp, dW1, db1, dW2, db2);
 *   swish = 1: x * sigmoid(x) instead of the ReLU (efficientnet-pytorch's MBConvBlock); C <= 4096, R <= 128.
 * vs_channel_gate / vs_channel_dot: x * g[n][c] over the map, and dg[n][c] = sum over positions of x * dy. */
int vs_pab_attention_fwd(int dtype, const void* top, const void* center, const void* bottom, const void* x, void* y, float* sp, float* scratch,
                         int n, int hw, int K, int C, void* stream);
int vs_pab_attention_bwd(int dtype, const void* dy, const void* top, const void* center, const void* bottom, const float* sp, void* dtop,
                         void* dcenter, void* dbottom, float* scratch, int n, int hw, int K, int C, void* stream);
size_t vs_pab_scratch_bytes(int n, int hw, int C);
int vs_se_gate_fwd(int dtype, const void* p, const float* w1, const float* b1, const float* w2, const float* b2, void* a, float* hid, int n,
                   int C, int R, int swish, void* stream);
int vs_se_gate_bwd(int dtype, const void* da, const void* a, const void* p, const float* hid, const float* w1, const float* w2, void* dp,
                   float* dw1, float* db1, float* dw2, float* db2, float* scratch, int n, int C, int R, int swish, void* stream);
size_t vs_se_gate_scratch_floats(int n, int C, int R);      /* floats of vs_se_gate_bwd's scratch */
int vs_channel_gate(int dtype, const void* x, const void* g, void* y, int n, int64_t hw, int c, void* stream);
int vs_channel_dot(int dtype, const void* x, const void* dy, void* dg, int n, int64_t hw, int c, void* stream);

/* ---- smp.PAN's Feature Pyramid Attention (decoders/pan/decoder.py: FPABlock) and the GAU gates, NHWC ----------------------------------
 * vs_maxpool2x2(_bwd): nn.MaxPool2d(2, 2) on all channels (the gradient goes to the first maximum in scan order).
 * vs_conv_to_plane(_bwd): a k x k convolution (padding k / 2, biased) from c channels to ONE: z fp32 [n][h][w]; w fp32 [k*k][c].
 * vs_fpa_pyramid_fwd/_bwd: everything between down1's convolution output z1 [n][h/2][w/2] (the caller writes it at the start of the
 *   arena) and the block's attention plane [n][h][w] - six single-channel ConvBnRelu layers (7x7, 5x5, 3x3), two max-pools, three
 *   bilinear (align_corners) upsamplings, in ONE workgroup; the arena (vs_fpa_arena_floats) keeps every intermediate for bwd, which
 *   leaves d z1 at arena[vs_fpa_dz1_offset] and the 6 x {dw, db, dgamma, dbeta} where `grads` points.  params / grads: HOST arrays
 *   of device pointers, 6 x {conv weight, conv bias, gamma, beta, running mean, running var} / 6 x {dw, db, dgamma, dbeta}; layer 0
 *   is the BatchNorm behind vs_conv_to_plane (its weight / bias slots are unused).
 * vs_fpa_combine(_bwd): out = plane * mid + b1 (b1 [n][c] broadcast).  vs_sigmoid(_bwd): GAU's gate.  vs_bn_fold_bias: shift +=
 *   scale * bias for a biased convolution in front of an evaluation-mode BatchNorm (smp's ConvBnRelu keeps the bias). */
int vs_maxpool2x2(int dtype, const void* x, void* y, int n, int h, int w, int c, void* stream);
int vs_maxpool2x2_bwd(int dtype, const void* x, const void* dy, void* dx, int n, int h, int w, int c, int accumulate, void* stream);
int vs_conv_to_plane(int dtype, const void* x, const float* w, const float* bias, float* z, int n, int h, int wd, int c, int k, void* stream);
int vs_conv_to_plane_bwd(int dtype, const void* x, const float* w, const float* dz, void* dx, float* dw, float* db, int n, int h, int wd, int c,
                         int k, void* stream);
size_t vs_fpa_arena_floats(int n, int h, int w);
size_t vs_fpa_dz1_offset(int n, int h, int w);
int vs_fpa_pyramid_fwd(float* arena, float* plane, float* const* params, int n, int h, int w, int training, void* stream);
int vs_fpa_pyramid_bwd(float* arena, const float* dplane, float* const* params, float* const* grads, int n, int h, int w, void* stream);
int vs_fpa_combine(int dtype, const float* plane, const void* mid, const void* b1, void* out, int n, int64_t hw, int c, void* stream);
int vs_fpa_combine_bwd(int dtype, const void* dy, const float* plane, const void* mid, void* dmid, float* dplane, int n, int64_t hw, int c, void* stream);
int vs_sigmoid(int dtype, const void* x, void* y, int64_t elems, void* stream);
int vs_sigmoid_bwd(int dtype, const void* dy, const void* y, void* dx, int64_t elems, void* stream);
int vs_bn_fold_bias(const float* scale, const float* bias, float* shift, int c, void* stream);

/* ---- RCCL behind the C ABI: the collectives of the two data-parallel splits on the caller's stream, one communicator per rank
 * (one process per GPU).  The reference has no multi-GPU path; volume-segmantics_amd/dist.py uses
 * torch.distributed by default and this transport with VOLSEG_COMM=rccl.  librccl is opened on first use (VS_ERR_UNSUPPORTED if it
 * is absent).  id: 128 bytes from rank 0's vs_comm_unique_id, handed to the other ranks by the launcher. */
typedef struct vs_comm vs_comm_t;
int vs_comm_unique_id(char id[128]);
int vs_comm_init(vs_comm_t** out, int nranks, int rank, const char id[128]);     /* on the current device */
void vs_comm_destroy(vs_comm_t* c);
int vs_comm_size(const vs_comm_t* c);
int vs_comm_rank(const vs_comm_t* c);
int vs_comm_allreduce_sum_f32(vs_comm_t* c, float* buf, int64_t n, void* stream);             /* gradients, in place */
int vs_comm_allreduce_max_u32(vs_comm_t* c, uint32_t* buf, int64_t n, void* stream);          /* packed keys: max = the merge */
int vs_comm_reduce_scatter_max_u32(vs_comm_t* c, const uint32_t* send, uint32_t* recv, int64_t n_per_rank, void* stream);
int vs_comm_allgather(vs_comm_t* c, const void* send, void* recv, int64_t bytes_per_rank, void* stream);
int vs_comm_broadcast(vs_comm_t* c, void* buf, int64_t bytes, int root, void* stream);
/* Diagnostics (bench.py's peak_crosscheck): TFLOP/s and mean in-kernel clock (GHz) of a register-only v_mfma_f32_16x16x32_bf16
 * loop on every CU, waves_per_simd 4-wave workgroups per CU - what the matrix pipes sustain on this box under its power
 * management, next to the 2.5 PFLOP/s of the data sheet. */
int vs_debug_mfma_rate(int iters, int waves_per_simd, double* tflops, double* clock_ghz);
/* The data-parallel form of the two shares: vs_unet_backward_part = the shares without the optimiser (role 2 = weight
 * gradients only); vs_unet_adamw_range = AdamW over the parameters of the units [unit_lo, unit_hi) from `grads` (after the
 * caller's all-reduce of that slice, vs_unet_unit_param_offset) plus their next-forward weight copies, in order on
 * `stream`.  Replaces the per-bucket hooks of a DistributedDataParallel wrap of the reference's loop
 * (vol_seg_2d_trainer.py:419-432); neither flips the weight set. */
int vs_unet_backward_part(vs_unet_t* net, const float* params, const float* x, const float* dlogits, int n,
                          int need_encoder_wgrad, float* grads, void* workspace, void* stream, int unit_lo, int unit_hi, int role);
int vs_unet_adamw_range(vs_unet_t* net, int need_encoder_wgrad, const float* grads, void* workspace, void* stream,
                        const vs_adamw_args* opt, int unit_lo, int unit_hi);
/* Dropout2d draws of a training forward (smp.FPN's decoder): mask = f(seed, *counter), counter = a device int64 the training
 * step advances (the engine passes encoder.bn1.num_batches_tracked), so a replayed graph draws a new mask every step.
 * vs_unet_dropout_mask_offset: byte offset in the training workspace of the last forward's mask ([n][channels] fp32), -1 if the
 * topology has no dropout (tests feed the mask to the oracle). */
int vs_unet_set_rng(vs_unet_t* net, uint32_t seed, const int64_t* counter);
int64_t vs_unet_dropout_mask_offset(const vs_unet_t* net);
/* the drop-connect draws of the last training forward (EfficientNet encoders): per MBConv block with a skip and a non-zero rate its block
 * index, rate and the byte offset in the training workspace of its [n] fp32 mask (0 or 1 / keep); returns their number */
int vs_unet_drop_connect_masks(const vs_unet_t* net, int64_t* offsets, int* blocks, float* rates, int cap);
/* the per-step scalars of a replayed optimiser step (see vs_adamw_args.hyper): one tiny launch on `stream` that also adds 1
 * to the n_bn BatchNorm num_batches_tracked counters (int64, may be NULL with n_bn = 0). */
int vs_train_hyper_set(float* hyper, float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                       int64_t* num_batches_tracked, int n_bn, void* stream);

/* DiceLoss(normalization="none") on raw logits and its gradient (data/pytorch3dunet_losses.py:15-41,89-135; the
 * trainer's default criterion, vol_seg_2d_trainer.py:133-135,425-428).  logits (n, K, h*w) fp32 NCHW, targets one-hot
 * (n, K, h*w) uint8 or fp32; loss: 1 device float; workspace (vs_dice_workspace bytes) carries the per-class sums from
 * the forward to the backward; grad_out: device scalar d(total)/d(loss) or NULL (= 1). */
size_t vs_dice_workspace(int classes);
int vs_dice_loss_fwd(const float* logits, const void* targets, int target_is_f32, int n, int classes, int64_t hw, float eps,
                     float* loss, float* workspace, size_t workspace_bytes, void* stream);
int vs_dice_loss_bwd(const float* logits, const void* targets, int target_is_f32, const float* grad_out, int n, int classes,
                     int64_t hw, float eps, const float* workspace, float* dlogits, void* stream);

/* The other criteria the trainer can select (vol_seg_2d_trainer.py:124-148), each as one reduction sweep + one gradient sweep over
 * logits (n, K, h*w) fp32 NCHW and one-hot targets (uint8 or fp32):
 *   kind 1: BCEDiceLoss(alpha, beta) = alpha * BCEWithLogitsLoss + beta * DiceLoss(normalization="sigmoid")  (pytorch3dunet_losses.py:171-184)
 *   kind 2: BCEWithLogitsLoss                 kind 3: CrossEntropyLoss (class index = position of the 1 in the one-hot column)
 *   kind 4: GeneralizedDiceLoss(normalization="sigmoid", epsilon = eps)  (pytorch3dunet_losses.py:138-169)
 * eps: the Dice / GDL clamp (1e-6 in the reference); workspace carries the gradient coefficients from fwd to bwd. */
size_t vs_seg_loss_workspace(int classes);
int vs_seg_loss_fwd(int kind, const float* logits, const void* targets, int target_is_f32, int n, int classes, int64_t hw,
                    float alpha, float beta, float eps, float* loss, float* workspace, size_t workspace_bytes, void* stream);
int vs_seg_loss_bwd(int kind, const float* logits, const void* targets, int target_is_f32, const float* grad_out, int n, int classes,
                    int64_t hw, const float* workspace, float* dlogits, void* stream);

/* MeanIoU, the trainer's default validation metric (data/pytorch3dunet_metrics.py:34-106; vol_seg_2d_trainer.py:150-161,
 * 243): per sample the prediction is the one-hot of the FIRST arg-max over channels (input > 0.5 for one channel), per
 * class |P & T| / max(|P | T|, 1e-8) with the target converted to bytes, mean over classes, then over samples.
 * input (n, K, h*w) fp32: probabilities, or with from_logits = 1 raw logits whose softmax(dim=1) is formed in fp32 inside the
 * kernel (the trainer's `softmax` + metric in one sweep; ties after rounding resolve as they would on the probabilities);
 * targets one-hot (n, K, h*w) uint8 or fp32; out: 1 device float; workspace: vs_mean_iou_workspace bytes (zeroed by the call). */
size_t vs_mean_iou_workspace(int n, int classes);
int vs_mean_iou(const float* input, const void* targets, int target_is_f32, int from_logits, int n, int classes, int64_t hw,
                float* out, void* workspace, size_t workspace_bytes, void* stream);

/* One-hot of a label batch on the device: labels (n, h*w) uint8 -> (n, K, h*w) uint8 (prepare_training_batch,
 * utilities/base_data_utils.py:150-158: mask -> one_hot(K) -> permute to NCHW -> uint8).  Labels >= K give an all-zero
 * column (torch.nn.functional.one_hot would raise). */
int vs_onehot_u8(const uint8_t* labels, int n, int classes, int64_t hw, uint8_t* onehot, void* stream);

/* ------------------------------------------------------------------------------------------
 * Training augmentations on the device (get_train_augs, data/augmentations.py:68-101; applied per sample by the reference's
 * DataLoader workers, data/datasets.py:42-60)
 * ---------------------------------------------------------------------------------------- */
/* One sample's draws, made by the host (every transform's coin and parameters; data/augmentations.py:sample_params). */
typedef struct vs_aug_params {
    int32_t crop, y1, x1, ch, cw;      /* RandomSizedCrop: window in the input image (crop = 0: off) */
    int32_t flip_v, rot_k, transpose;  /* VerticalFlip, RandomRotate90 (k quarter turns, counter-clockwise), Transpose */
    int32_t distort;                   /* OneOf: 0 none, 1 ElasticTransform, 2 GridDistortion, 3 OpticalDistortion */
    float inv_affine[6];               /* elastic: the sampling (inverse) affine map, row major 2 x 3 */
    float k, cx, cy;                   /* optical: distortion coefficient, principal point */
    uint32_t noise_seed;               /* elastic: seed of the displacement noise */
    float clahe_clip;                  /* CLAHE clip limit (0 = off) */
    int32_t clahe_limit;               /* max(int(clip * tile_area / 256), 1) */
} vs_aug_params;
/* images / masks: n x size x size uint8 (resident); params_dev: n structs; luts_dev: n x 256 uint8 intensity maps
 * (RandomBrightnessContrast / RandomGamma / identity); grid_tables_dev: n x 2 x size floats (GridDistortion's x / y coordinate
 * tables).  out_x: (n, 1, size, size) fp32 normalised network input, out_masks: n x size x size uint8.  fields_out (optional,
 * n x 2 x size x size floats): the elastic displacement fields that were used (tests).  size: a multiple of 8. */
size_t vs_augment_workspace(int n, int size);
int vs_augment_batch(const uint8_t* images, const uint8_t* masks, int n, int size, const vs_aug_params* params_dev,
                     const uint8_t* luts_dev, const float* grid_tables_dev, float* out_x, uint8_t* out_masks,
                     void* workspace, size_t workspace_bytes, float* fields_out, void* stream);

/* ------------------------------------------------------------------------------------------
 * Volume pre-processing (BaseDataManager._preprocess_data, data/base_data_manager.py:29-42;
 * clip_to_uint8, utilities/base_data_utils.py:243-287)
 * ---------------------------------------------------------------------------------------- */
enum { VS_VOL_F32 = 0, VS_VOL_F64 = 1, VS_VOL_U8 = 2, VS_VOL_I8 = 3, VS_VOL_U16 = 4, VS_VOL_I16 = 5, VS_VOL_U32 = 6, VS_VOL_I32 = 7,
       VS_VOL_I64 = 8, VS_VOL_U64 = 9 };

/* One reduction pass over a C-contiguous volume of n elements, in NumPy's add.reduce order (8192-element buffers added in
 * sequence, pairwise summation inside a buffer) and accumulation type (float for VS_VOL_F32, double otherwise), so that the
 * np.nanmean / np.nanstd the reference takes (base_data_manager.py:33, base_data_utils.py:255) can be reproduced bit for bit:
 *   op 0: out[0] = sum of x with NaN -> 0, out[1] = number of NaNs;
 *   op 1: out[0] = sum of (x - avg)^2 with NaN -> 0 (avg rounded to the accumulation type).
 * out: 2 doubles on the device.  workspace: vs_volume_sum_workspace(n) bytes on the device. */
size_t vs_volume_sum_workspace(int64_t n);
int vs_volume_sum(int vtype, const void* data, int64_t n, int op, double avg, void* workspace, size_t workspace_bytes,
                  double* out, void* stream);

/* 2 x 2 x 2 block mean of an INTEGER volume [d][h][w] into float64 [(d+1)/2][(h+1)/2][(w+1)/2], odd edges zero padded and still
 * divided by 8: `downsample_data` = skimage.measure.block_reduce(data, (2, 2, 2), np.nanmean) - utilities/base_data_utils.py:161-163
 * (called from data/base_data_manager.py:36-38 when `downsample: True`).  Exact (block sums of integers are exact in float64). */
int vs_downsample2x_mean(int vtype, const void* data, double* out, int d, int h, int w, void* stream);

/* out[i] = uint8(clip((clip(x, lower, upper) - lower) / (upper - lower), 0, 1) * 255), NaN -> nan_fill first, every step
 * rounded in the volume's float type (float32 volumes) or in float64 (float64 and integer volumes) as NumPy's in-place
 * ufuncs do (base_data_utils.py:270-287).  counts (optional, 2 x uint64 on the device, accumulated): voxels above upper /
 * below lower (the numbers the reference logs, :259-268). */
int vs_clip_to_uint8(int vtype, const void* data, int64_t n, double nan_fill, double lower, double upper, uint8_t* out,
                     uint64_t* counts, void* stream);

/* ------------------------------------------------------------------------------------------
 * Prediction path (vol_seg_2d_predictor.py:31-136)
 * ---------------------------------------------------------------------------------------- */
/* Index map of one prediction direction: voxel address of slice s, row h, column w of the
 * rotated/swapped stack = base + s*ss + h*sh + w*sw (elements).  np.rot90 / swapaxes
 * (vol_seg_2d_predictor.py:34,108; base_data_utils.py:132-138) are pure index maps. */
typedef struct vs_dirmap {
    int64_t base, ss, sh, sw;
    int32_t depth, h, w;      /* stack dims (un-padded) */
    int32_t hp, wp;           /* padded to multiples of 32 (augmentations.py:30-65) */
    int32_t pad_top, pad_left;   /* PadIfNeeded centre offsets */
    int32_t crop_top, crop_left; /* torchvision center_crop offsets (base_data_utils.py:125-129) */
} vs_dirmap;

/* x[b][hp][wp] fp32 = normalised, reflect-101 padded slices s0..s0+nb-1 of the uint8 volume
 * (data/datasets.py:120-142: /255, -0.449, /0.226). */
int vs_slices_gather(const uint8_t* vol, const vs_dirmap* m, int s0, int nb, float* x, void* stream);
/* the same for a volume of any VS_VOL_* type, in the arithmetic numpy uses at data/datasets.py:128-134: integer types ->
 * float32, / 255 (whatever their range: an unclipped uint16 volume gives inputs >> 1, as in the reference), float32 volumes
 * skip the / 255, float64 volumes are normalised in float64 and rounded to float32 at the end. */
int vs_slices_gather_typed(int vtype, const void* vol, const vs_dirmap* m, int s0, int nb, float* x, void* stream);

/* softmax -> argmax (first max) -> max prob (fp32 -> fp16 RNE) on logits (nb, K, hp, wp) NCHW,
 * centre-cropped and scattered to voxel addresses (vol_seg_2d_predictor.py:45-64).
 * mode 0: labels[u8] and probs[f16] volumes (either may be null);
 * mode 1: keys[addr] = max(keys[addr], prob_bits<<16 | (15-dir)<<8 | label)  (packed-key merge);
 * mode 2: votes[label][addr] += 1 (one-hot variants, :118-136); mode 3: += 2 - a direction of the 12-way scheme (:100-116) that a later
 * one repeats exactly (same slices, same voxel addresses: the later one is then not run). */
int vs_logits_to_volume(const float* logits, int classes, const vs_dirmap* m, int s0, int nb, int mode,
                        int direction, uint8_t* labels, uint16_t* probs, uint32_t* keys, uint8_t* votes,
                        int64_t nvox, void* stream);

/* The two calls above in one: eval-mode forward of the slices x (nb, 1, hp, wp) whose segmentation head goes straight
 * to the volume(s) - with up to 4 classes and modes 0 / 1 no logits are written at all (the head kernel's epilogue does the
 * softmax / arg-max / crop / scatter; keys through an order-free atomic max); otherwise the logits land in the plan's
 * workspace and vs_logits_to_volume runs on them.  Results are identical to vs_unet_forward + vs_logits_to_volume. */
int vs_unet_forward_to_volume(vs_unet_t* net, const float* params, float* bnstate, const float* x, int nb, void* workspace,
                              void* stream, const vs_dirmap* m, int s0, int mode, int direction, uint8_t* labels,
                              uint16_t* probs, uint32_t* keys, uint8_t* votes, int64_t nvox);

/* The reference's pairwise merge (_merge_vols_in_mem, :90-98): where prob1 > prob0 take slot 1. */
int vs_merge_maxprob(uint8_t* label0, uint16_t* prob0, const uint8_t* label1, const uint16_t* prob1,
                     int64_t n, void* stream);
int vs_keys_unpack(const uint32_t* keys, uint8_t* labels, uint16_t* probs, int64_t n, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VOLSEG_HIP_H */
