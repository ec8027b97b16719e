// Whole-network plan for smp.Unet(encoder_name="resnet34", in_channels=1, classes=K)
// (reference construction: volume_segmantics/model/model_2d.py:15-16; topology: SURVEY.md section 8a).
//
// The plan is a flat list of units (stem, maxpool, conv+BN(+residual)(+ReLU), head) with all activation,
// gradient and scratch buffers laid out once in a caller-owned workspace.  vs_unet_forward /
// vs_unet_backward enqueue every kernel of a step from C++ on the caller's stream - Python makes one
// call per pass.  Forward in training mode keeps the pre-BN (z) and post-activation (a) tensors;
// backward walks the units in reverse with a statically known first-write / accumulate discipline for
// the gradients of tensors with two consumers (ResNet identities, U-Net skips).
#include <cstdlib>
#include <cmath>
#include <string>
#include <vector>

#include "common.h"
#include "prof.h"

int launch_weight_prepare(int dtype, const float* w, void* wc, void* wt, int cout, int taps, int cin, int cout_pad, hipStream_t s);
int launch_dlogits_to_nhwc16(int dtype, const float* d, void* o, int n, int k, int64_t hw, float* db, float* partial, hipStream_t s);
extern "C" int vs_depth_to_space2(int dtype, const void* x, void* y, int n, int h, int w, int c, const float* bias, const float* scale,
                                  const float* shift, int relu, void* stream);
extern "C" int vs_space_to_depth2(int dtype, const void* x, void* y, int n, int h, int w, int c, void* stream);
extern "C" int vs_colsum(int dtype, const void* x, int64_t rows, int c, float* out, float* workspace, size_t workspace_bytes, void* stream);
extern "C" int vs_upsample2x_add(int dtype, const void* x, const void* skip, void* y, int n, int h, int w, int c, void* stream);
extern "C" size_t vs_gn_bwd_workspace(int n, int c, int groups);
extern "C" int vs_gn_fwd(int dtype, const void* x, const float* gamma, const float* beta, int relu, void* y, float* stats, int n, int64_t hw,
                         int c, int groups, float eps, float* workspace, size_t workspace_bytes, void* stream);
extern "C" int vs_gn_bwd(int dtype, const void* dy, const void* x, const float* stats, const float* gamma, const float* beta, int relu,
                         void* dx, float* dgamma, float* dbeta, int n, int64_t hw, int c, int groups, float* workspace,
                         size_t workspace_bytes, void* stream);
extern "C" int vs_bilinear_up(int dtype, const void* x, void* y, int n, int h, int w, int c, int factor, void* stream);
extern "C" int vs_bilinear_up_bwd(int dtype, const void* dy, void* dx, int n, int h, int w, int c, int factor, int accumulate, void* stream);
extern "C" int vs_bilinear_up_planes(const float* x, float* y, int planes, int h, int w, int factor, void* stream);
extern "C" int vs_bilinear_up_planes_bwd(const float* dy, float* dx, int planes, int h, int w, int factor, void* stream);
extern "C" int vs_dropout2d_mask(float* mask, int n, int c, float p, uint32_t seed, const int64_t* counter, int64_t bias, void* stream);
extern "C" int vs_channel_scale(int dtype, const void* x, const float* mask, void* y, int n, int64_t hw, int c, void* stream);
extern "C" int vs_dwconv3x3(int dtype, const void* x, const float* w, void* y, int n, int h, int wd, int c, int dilation, int flip, void* stream);
extern "C" size_t vs_dwconv3x3_wgrad_workspace(int c);
extern "C" int vs_dwconv3x3_wgrad(int dtype, const void* x, const void* dy, float* dw, int n, int h, int wd, int c, int dilation, float* workspace,
                                  size_t workspace_bytes, void* stream);
extern "C" int vs_spatial_sum(int dtype, const void* x, void* y, int n, int64_t hw, int c, float scale, void* stream);
extern "C" int vs_broadcast_rows(int dtype, const void* v, void* y, int n, int64_t hw, int c, float scale, int accumulate, void* stream);
extern "C" int vs_dropout(int dtype, const void* x, void* y, int64_t elems, float p, uint32_t seed, const int64_t* counter, int64_t bias, void* stream);
extern "C" int vs_dilated_im2col(int dtype, const void* src, void* dst, int n, int h, int w, int c, int r, int inverse, int accumulate, void* stream);
extern "C" int vs_pab_attention_fwd(int dtype, const void* top, const void* center, const void* bottom, const void* x, void* y, float* sp,
                                    float* scratch, int n, int hw, int K, int C, void* stream);
extern "C" int vs_pab_attention_bwd(int dtype, const void* dy, const void* top, const void* center, const void* bottom, const float* sp, void* dtop,
                                    void* dcenter, void* dbottom, float* scratch, int n, int hw, int K, int C, void* stream);
extern "C" size_t vs_pab_scratch_bytes(int n, int hw, int C);
extern "C" int vs_se_gate_fwd(int dtype, const void* p, const float* w1, const float* b1, const float* w2, const float* b2, void* a, float* hid,
                              int n, int C, int R, int swish, void* stream);
extern "C" int vs_se_gate_bwd(int dtype, const void* da, const void* a, const void* p, const float* hid, const float* w1, const float* w2, void* dp,
                              float* dw1, float* db1, float* dw2, float* db2, float* scratch, int n, int C, int R, int swish, void* stream);
extern "C" size_t vs_se_gate_scratch_floats(int n, int C, int R);
extern "C" int vs_channel_gate(int dtype, const void* x, const void* g, void* y, int n, int64_t hw, int c, void* stream);
extern "C" int vs_channel_dot(int dtype, const void* x, const void* dy, void* dg, int n, int64_t hw, int c, void* stream);
extern "C" int vs_maxpool2x2(int dtype, const void* x, void* y, int n, int h, int w, int c, void* stream);
extern "C" int vs_maxpool2x2_bwd(int dtype, const void* x, const void* dy, void* dx, int n, int h, int w, int c, int accumulate, void* stream);
extern "C" int vs_conv_to_plane(int dtype, const void* x, const float* w, const float* bias, float* z, int n, int h, int wd, int c, int k, void* stream);
extern "C" int vs_conv_to_plane_bwd(int dtype, const void* x, const float* w, const float* dz, void* dx, float* dw, float* db, int n, int h, int wd,
                                    int c, int k, void* stream);
extern "C" size_t vs_fpa_arena_floats(int n, int h, int w);
extern "C" size_t vs_fpa_dz1_offset(int n, int h, int w);
extern "C" int vs_fpa_pyramid_fwd(float* arena, float* plane, float* const* params, int n, int h, int w, int training, void* stream);
extern "C" int vs_fpa_pyramid_bwd(float* arena, const float* dplane, float* const* params, float* const* grads, int n, int h, int w, void* stream);
extern "C" int vs_fpa_combine(int dtype, const float* plane, const void* mid, const void* b1, void* out, int n, int64_t hw, int c, void* stream);
extern "C" int vs_fpa_combine_bwd(int dtype, const void* dy, const float* plane, const void* mid, void* dmid, float* dplane, int n, int64_t hw, int c,
                                  void* stream);
extern "C" int vs_sigmoid(int dtype, const void* x, void* y, int64_t elems, void* stream);
extern "C" int vs_sigmoid_bwd(int dtype, const void* dy, const void* y, void* dx, int64_t elems, void* stream);
extern "C" int vs_bn_fold_bias(const float* scale, const float* bias, float* shift, int c, void* stream);
int launch_adamw_prepare_all(int dtype, const vs_adamw_args& a, const float* grads, void* ws, int n, const long* w_off, const long* wc_off,
                             const long* wt_off, const int* cout, const int* taps, const int* cin, const int* cout_pad, const int* cg,
                             const int* update, hipStream_t s);
int launch_weight_prepare_all(int dtype, const float* params, void* ws, int n, const long* w_off, const long* wc_off,
                              const long* wt_off, const int* cout, const int* taps, const int* cin, const int* cout_pad,
                              const int* cg, hipStream_t s);

namespace {

struct TensorInfo {
    std::string name;
    int64_t shape[4];
    int ndim, kind;
    int64_t offset;
};

enum UnitKind { U_STEM, U_POOL, U_CONV, U_HEAD, U_CONCAT,
                U_CONVT,   // ConvTranspose2d(4, stride 2, padding 1) + BN + ReLU (smp Linknet's TransposeX2): a 3x3 convolution onto
                           // 4 * cout channels (conv_igemm) + pixel shuffle; cin0 -> cout channels, hin x win -> hout x wout = 2x
                U_ADD,     // out = a(src0) + a(src1) (Linknet's skip connection, FPN's merge)
                U_UPADD,   // out = nearest-x2 upsampling of a(src0) + a(src1) (smp FPNBlock)
                U_BILINEAR,  // out = bilinear x2 upsampling (align_corners) of a(src0) (smp Conv3x3GNReLU(upsample=True))
                U_DROPOUT,   // nn.Dropout2d(0.2) in training, the identity in evaluation (smp FPNDecoder.dropout)
                U_DWCONV,    // depthwise 3x3 convolution, dilation = padding = dil (first half of smp's SeparableConv2d), no norm
                U_GAP,       // nn.AdaptiveAvgPool2d(1): out [n][1][1][c]
                U_BCAST,     // F.interpolate of a 1x1 map to hout x wout (ASPPPooling)
                U_DROPOUT_E,    // element-wise nn.Dropout(0.5) (ASPP.project), the identity in evaluation
                U_PAB,          // smp MAnet's PAB attention: out = src0 + reinterpret(softmax_all(center top^T) bottom); members = {top, center, bottom}
                U_SE,           // squeeze-excitation gate on a pooled [n][1][1][c] feature: tensors w_idx .. w_idx + 3 = W1, b1, W2, b2; cin1 = hidden width
                U_CGATE,        // out = a(src0) * gate a(src1) ([n][1][1][c]) over the map
                U_SIGMOID,      // element-wise sigmoid (smp PAN's GAU gate)
                U_BN,           // standalone nn.BatchNorm2d(cout, bn_eps, bn_mom) + activation `relu` (0 none / 1 ReLU / 2 swish) on a(src0), any
                                // channel count (csrc/effnet.hip): the norms of smp's EfficientNet encoders
                U_DWCONV2,      // depthwise k x k convolution, stride 1 / 2, `pad` zero rows / columns in front (static same padding), no
                                // norm; bcast: the network input (one fp32 channel) broadcast over cout = the EfficientNet stem
                U_DROPADD,      // out = drop_connect(a(src0), drop_p) + a(src1) (the MBConv skip); evaluation: the plain sum
                U_FOLD2,        // out [.., c] = in [.., :c] + in [.., c:] (the sum over ResNeSt's two radix splits; maps and pooled vectors)
                U_RSOFTMAX,     // timm's RadixSoftmax(2, 1) on attention logits [n][2 c]
                U_RADIXSUM,     // out [.., c] = a(src0)[.., :c] * gate[:c] + a(src0)[.., c:] * gate[c:], gate = a(src1) [n][2 c] (the attention-weighted
                                // sum of ResNeSt's two splits in one sweep)
                U_AVGPOOL,      // nn.AvgPool2d(k, stride) of ResNeSt's avd (3, padding 1, zeros counted) / avg_down (2): a depthwise convolution
                                // with constant taps pool_w
                U_UP2,          // out = nearest-x2 upsampling of a(src0), materialised - where a decoder's upsample + concat cannot ride the
                                // convolution's loader (the boundary is not a multiple of 32 channels: EfficientNet features of 136 / 56 / 48)
                U_FPA };        // smp PAN's FPABlock pyramid + combination: out = plane(src0) * a(src1) + a(res) broadcast; tens = its 24
                                // parameter tensors (6 x conv weight, conv bias, BN gamma, BN beta)

struct Act {  // one activation tensor (per-sample element count = c*h*w)
    int c, h, w;
    bool has_z;
    size_t off_a = 0, off_z = 0, off_da = 0, off_dz = 0;
};

// rows of fixed-point statistics bins of a unit (ConvParams::stats_bins): enough rows that an address takes ~16 - 32 atomic adds per
// launch (16 rows for the tile kernels' 256 - 512 tiles; 256 for the <= 16-channel layers, whose direct kernel adds once per WAVE)
static inline int stat_bins_rows(int cout) { return cout <= 16 ? 256 : 16; }
static inline size_t unit_bins_bytes(int cout) { return (size_t)stat_bins_rows(cout) * 2 * cout * sizeof(unsigned long long); }
constexpr size_t kTicketBytes = 128;   // behind a unit's bins: the ticket counter of ConvParams::fin_ticket (cleared with the bins)

struct Unit {
    UnitKind kind;
    int src0 = -1, src1 = -1, up0 = 0;  // input activation ids
    int cin0 = 0, cin1 = 0, cout = 0, k = 3, stride = 1, pad = 1;
    int hin = 0, win = 0, hout = 0, wout = 0;  // virtual input / output spatial dims
    int w_idx = -1, bn_idx = -1, bias_idx = -1;  // indices into the tensor table (bn_idx -> gamma)
    int out = -1;   // output activation id
    int res = -1;   // residual activation id
    int relu = 1;
    int gn_idx = -1, gn_groups = 0;   // U_CONV followed by nn.GroupNorm(gn_groups, cout) + ReLU instead of BatchNorm (gamma at gn_idx)
    size_t off_gn = 0;                // its statistics [n][groups][2] fp32
    float bn_eps = 1e-5f, bn_mom = 0.1f;   // U_BN
    float drop_p = 0.f; int salt = 0;      // U_DROPADD: drop-connect rate, block index (separates the blocks' draws)
    int bcast = 0;                         // U_DWCONV2 on the single-channel network input
    int g2 = 0;                            // U_CONV: cin -> cout = 2 cin in TWO groups (ResNeSt's radix-2 split-attention 3x3); weights [cout][taps][cin / 2],
                                           // the compute copies dense with the other group's half zero (optim.hip: two_groups)
    bool aux_frozen = false;               // the unit's BatchNorm / bias tensors also match the freeze predicate ("encoder" and "conv" in the name)
    float pool_w = 0.f;                    // U_AVGPOOL: the constant tap (1 / 9 or 1 / 4)
    int dil = 1;    // dilation of a stride-1 3x3 convolution (2: smp's replace_strides_with_dilation; any for U_DWCONV)
    int factor = 2; // U_BILINEAR: integer scale factor
    int colr = 0;   // U_CONV: a 3x3 convolution with dilation = padding = colr (DeepLabV3's dense ASPP rates 12 / 24 / 36), run as the
                    // 1x1 convolution over the 9 * cin channels of its input's column form (vs_dilated_im2col)
    size_t off_xs = 0;   // that column form (kept for the weight gradient)
    int cg = 0;     // grouped convolution (ResNeXt): channels per group, cin0 == cout; 0 = dense.  Weights [cout][k*k][cg]; the
                    // compute copies are block-expanded to 32-channel super-groups (vs_weights_prepare_grouped)
    bool frozen_candidate = false;  // "encoder" in name and "conv" in name (vol_seg_2d_trainer.py:102-108)
    size_t off_wc = 0, off_wt = 0, off_bn = 0;  // workspace offsets (bytes): weight copies, 4*C floats of BN constants
    size_t off_bins = 0;                         // stat_bins_rows(cout) rows of fixed-point statistics bins ([row][2][cout] 64-bit), training plans
    size_t off_wc2 = 0, off_wt2 = 0;             // second set of weight copies (training workspaces): see vs_unet::wset
    std::vector<int> tens;                       // U_FPA: parameter tensor indices
    size_t off_fpa_pool = 0, off_fpa_arena = 0, off_fpa_plane = 0;   // U_FPA: pooled input, pyramid arena (fp32), attention plane (fp32)
    std::vector<int> members;                    // U_CONCAT: the activations whose channels `out` strings together, in order
};

struct Layout {
    std::vector<TensorInfo> tensors;
    int64_t n_params = 0, n_bnstate = 0;
};

void add_tensor(Layout& L, const std::string& name, std::initializer_list<int64_t> shape, int kind) {
    TensorInfo t;
    t.name = name;
    t.ndim = (int)shape.size();
    int64_t numel = 1;
    int i = 0;
    for (auto s : shape) { t.shape[i++] = s; numel *= s; }
    for (; i < 4; ++i) t.shape[i] = 1;
    t.kind = kind;
    if (kind <= 3) { t.offset = L.n_params; L.n_params += numel; }
    else { t.offset = L.n_bnstate; L.n_bnstate += numel; }
    L.tensors.push_back(t);
}

// returns index of gamma
int add_bn(Layout& L, const std::string& prefix, int c) {
    const int idx = (int)L.tensors.size();
    add_tensor(L, prefix + ".weight", {c}, 1);
    add_tensor(L, prefix + ".bias", {c}, 2);
    add_tensor(L, prefix + ".running_mean", {c}, 4);
    add_tensor(L, prefix + ".running_var", {c}, 5);
    return idx;
}

}  // namespace

struct vs_unet {
    int dtype, classes, max_batch, h, w;
    int topology = 0;   // 0 = smp.Unet, 1 = smp.UnetPlusPlus (dense nested skips)
    int encoder = 34;   // torchvision ResNet depth behind smp's encoder_name: 18 / 34 (BasicBlock) or 50 (Bottleneck)
    Layout layout;
    std::vector<Act> acts;
    std::vector<Unit> units;
    size_t esz;
    // workspace regions (bytes)
    size_t off_bins0 = 0, bins_bytes = 0;        // all units' statistics bins, contiguous: zeroed by ONE launch per training forward
    size_t off_bnws = 0, bnws_bytes = 0, off_wgws = 0, wgws_bytes = 0, off_headdw = 0, off_headpart = 0, off_dyh = 0, off_dup = 0,
           off_zs = 0, off_idx = 0;
    size_t ws_eval = 0, ws_train = 0, off_logits = 0, off_bncnt = 0;
    int last_n = 0;
    std::vector<char> group_first;  // optimiser groups of the fused backward (see unet_backward_range)
    std::vector<int> producer, first_consumer;   // per activation: unit that outputs it / lowest-index unit that reads it
    std::vector<int> bwd_stat_rows;              // per activation: partial rows left by the dgrad that completed its gradient
    std::vector<int> sole_consumer;              // per activation: the ONE unit that reads it (-1: none / several readers)
    // SyncBatchNorm under data parallelism (vs_unet_set_stats_hook): the statistics of every BatchNorm are summed over the ranks
    vs_stats_hook_fn stats_hook = nullptr; void* stats_user = nullptr; int stats_world = 1;
    size_t off_syncsc = 0;                       // 2 * cmax floats: the summed copies of a unit's (dbeta, dgamma) for the backward apply
    std::vector<char> nl_act;                    // per activation, set by the last training forward: it was never materialised - its one
                                                 // consumer normalises the producer's pre-norm output z while loading it (ConvParams::nl_*)
    size_t off_gnz = 0, off_gnws = 0, gnws_bytes = 0, off_dropmask = 0, off_lsmall = 0, off_dlsmall = 0;   // smp.FPN (see plan_workspace)
    int head_up = 1;                   // the head works at 1 / head_up resolution, nn.UpsamplingBilinear2d(head_up) follows (FPN: 4)
    uint32_t rng_seed = 0; const int64_t* rng_counter = nullptr;   // Dropout2d draws (vs_unet_set_rng)
    size_t off_pab = 0, pab_bytes = 0; // scratch of the PAB attention (vs_pab_scratch_bytes)
    size_t off_sews = 0;               // scratch of the squeeze-excitation gates' backward pass (vs_se_gate_scratch_floats)
    size_t off_gapws = 0, gapws_bytes = 0;   // split partial sums of the average pools / gate gradients (vs_sample_rowsum_ws)
    size_t off_avgw9 = 0, off_avgw4 = 0; int avgw_c = 0;   // constant taps of U_AVGPOOL: [c][9] of 1 / 9 and [c][4] of 1 / 4
    size_t off_ys = 0;                 // scratch: the column form of a large-rate convolution's input gradient
    size_t off_ct = 0, off_ctdw = 0, ctdw_bytes = 0;   // transposed convolutions: un-shuffled output; dense 3x3 weight gradient
    int wset = 0;  // which set of weight copies the forward / backward read; the fused optimiser step fills the other and flips
    std::vector<char> written;  // per activation: has its gradient buffer been written in the current backward pass
    // backward runs the weight-gradient kernels on an internal side stream, forked from / joined to the caller's stream
    static constexpr int kSide = 2;
    hipStream_t side[kSide] = {nullptr, nullptr};
    std::vector<hipEvent_t> fork_events;
    hipEvent_t join_event[kSide] = {nullptr, nullptr};
    ~vs_unet() {
        for (auto e : fork_events) (void)hipEventDestroy(e);
        for (int i = 0; i < kSide; ++i)
            if (join_event[i]) (void)hipEventDestroy(join_event[i]);   // (the side streams belong to the process-wide pool below)
    }
};

// ---- the side streams: one pair per device for every plan of the process, checked against the caller's stream -------------------
// HIP multiplexes its streams onto a few hardware queues (GPU_MAX_HW_QUEUES, 4 by default).  A side stream that shares its queue with
// the caller's stream runs its kernels IN ORDER with the caller's: the weight gradients no longer overlap the backward pass and the
// batch-32 step takes 5.3 instead of 4.4 ms - measured for every second model a process creates when each plan made its own streams
// (tools/placement_probe.py, profiles/r4_side_stream_hw_queues.txt).  So the streams are made once, and the first time they serve a
// given caller's stream a 2 x 100 us timing probe (one spin kernel on each stream, started together) tells whether the two really run
// beside one another; a stream that does not is parked (kept, so that the runtime's assignment moves on) and another one is tried.
__global__ void side_probe_spin(unsigned long long ticks) {   // ticks of the 100 MHz wall clock
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}
struct SidePool {
    hipStream_t s[vs_unet::kSide] = {nullptr, nullptr};
    hipStream_t checked_for = nullptr;
    bool checked = false;
    std::vector<hipStream_t> parked;
};
static SidePool& side_pool() {
    static SidePool pools[64];
    int dev = 0;
    (void)hipGetDevice(&dev);
    return pools[dev & 63];
}
// true: a kernel on `a` and a kernel on `b` ran at the same time
static int streams_overlap(hipStream_t a, hipStream_t b, bool* yes) {
    hipEvent_t e0, e1, eb;
    VS_CHECK_HIP(hipEventCreate(&e0));
    VS_CHECK_HIP(hipEventCreate(&e1));
    VS_CHECK_HIP(hipEventCreateWithFlags(&eb, hipEventDisableTiming));
    VS_CHECK_HIP(hipStreamSynchronize(a));
    VS_CHECK_HIP(hipStreamSynchronize(b));
    float best = 1e30f;
    for (int rep = 0; rep < 2; ++rep) {       // (the first pass also pays the kernel's load)
        VS_CHECK_HIP(hipEventRecord(e0, a));
        hipLaunchKernelGGL(side_probe_spin, dim3(1), dim3(64), 0, a, 10000ull);
        hipLaunchKernelGGL(side_probe_spin, dim3(1), dim3(64), 0, b, 10000ull);
        VS_CHECK_HIP(hipEventRecord(eb, b));
        VS_CHECK_HIP(hipStreamWaitEvent(a, eb, 0));
        VS_CHECK_HIP(hipEventRecord(e1, a));
        VS_CHECK_HIP(hipStreamSynchronize(a));
        float ms = 0.f;
        VS_CHECK_HIP(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, ms);
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipEventDestroy(eb);
    *yes = best < 0.160f;                     // two 100 us kernels: ~0.11 ms beside one another, ~0.21 ms one after the other
    return VS_OK;
}
static int acquire_side_streams(vs_unet* net, hipStream_t caller) {
    SidePool& pool = side_pool();
    for (int i = 0; i < vs_unet::kSide; ++i)
        if (!pool.s[i]) VS_CHECK_HIP(hipStreamCreateWithFlags(&pool.s[i], hipStreamNonBlocking));
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(caller, &cap);
    if (cap == hipStreamCaptureStatusNone && !(pool.checked && pool.checked_for == caller)) {
        for (int i = 0; i < vs_unet::kSide; ++i) {
            for (int attempt = 0; attempt < 8; ++attempt) {
                bool ok = false;
                const int rc = streams_overlap(caller, pool.s[i], &ok);
                if (rc) return rc;
                if (ok && i == 1) { const int rc2 = streams_overlap(pool.s[0], pool.s[1], &ok); if (rc2) return rc2; }
                if (ok) break;
                pool.parked.push_back(pool.s[i]);
                VS_CHECK_HIP(hipStreamCreateWithFlags(&pool.s[i], hipStreamNonBlocking));
            }
        }
        pool.checked = true;
        pool.checked_for = caller;
    }
    for (int i = 0; i < vs_unet::kSide; ++i) net->side[i] = pool.s[i];
    return VS_OK;
}

namespace {

int build(vs_unet* net) {
    Layout& L = net->layout;
    auto& A = net->acts;
    auto& U = net->units;
    const int H = net->h, W = net->w;
    auto new_act = [&](int c, int h, int w, bool has_z) {
        Act a; a.c = c; a.h = h; a.w = w; a.has_z = has_z;
        A.push_back(a);
        return (int)A.size() - 1;
    };
    // cat([nearest-x2(x), skip members...]) as ONE activation (U_UP2 + U_CONCAT): the route for boundaries the fused loader does not take
    auto materialised_cat = [&](int x_act, int x_c, const std::vector<int>& skip_members, int skip_c) {
        const Act xa = A[x_act];
        Unit up; up.kind = U_UP2; up.src0 = x_act; up.cout = x_c; up.hin = xa.h; up.win = xa.w; up.hout = 2 * xa.h; up.wout = 2 * xa.w; up.relu = 0;
        up.out = new_act(x_c, 2 * xa.h, 2 * xa.w, false);
        U.push_back(up);
        Unit cu; cu.kind = U_CONCAT; cu.members = {up.out};
        for (int m : skip_members) cu.members.push_back(m);
        cu.cout = x_c + skip_c; cu.hout = 2 * xa.h; cu.wout = 2 * xa.w; cu.relu = 0;
        cu.out = new_act(x_c + skip_c, 2 * xa.h, 2 * xa.w, false);
        U.push_back(cu);
        return cu.out;
    };
    // ---- encoder ----
    int feat[6]; int featc[6] = {0, 64, 0, 0, 0, 0};          // activations / channels of the encoder features the decoder taps
    int cur = -1, inpl = 64, ch = H / 4, cw = W / 4;
    if (net->encoder == 103 || net->encoder == 104) {
        // smp's EfficientNetEncoder (encoders/efficientnet.py) over efficientnet-pytorch 0.6.3's EfficientNet-b3 / b4: stem 3x3 / 2 + BN +
        // swish; MBConv blocks = [expand 1x1 + BN + swish] depthwise k x k + BN + swish, squeeze-excitation (hidden = max(1, input
        // filters / 4), swish), project 1x1 + BN, drop_connect(0.2 * i / blocks) + skip when the shape is kept; BatchNorm2d(momentum 0.01,
        // eps 1e-3); Conv2dStaticSamePadding at the (even) nominal image size: (0, 1) for k = 3 and (1, 2) for k = 5 at stride 2.
        // _conv_head / _bn1 stay in the state dict and are never run.  Registration order = execution order.
        const bool b4 = net->encoder == 104;
        const double width = b4 ? 1.4 : 1.2, depth = b4 ? 1.8 : 1.4;
        auto round_filters = [&](int f) { const double x = f * width; int nw = std::max(8, (int)(x + 4) / 8 * 8); if (nw < 0.9 * x) nw += 8; return nw; };
        auto round_repeats = [&](int r) { return (int)std::ceil(depth * r); };
        static const int base[7][6] = {{1, 3, 1, 1, 32, 16}, {2, 3, 2, 6, 16, 24}, {2, 5, 2, 6, 24, 40}, {3, 3, 2, 6, 40, 80}, {3, 5, 1, 6, 80, 112},
                                       {4, 5, 2, 6, 112, 192}, {1, 3, 1, 6, 192, 320}};
        static const int ends_b3[4] = {5, 8, 18, 26}, ends_b4[4] = {6, 10, 22, 32};
        const int* ends = b4 ? ends_b4 : ends_b3;
        auto bn_unit = [&](const std::string& name, int src, int cch, int hh, int ww, int act) {
            Unit u; u.kind = U_BN; u.src0 = src; u.cin0 = cch; u.cout = cch; u.hin = u.hout = hh; u.win = u.wout = ww; u.relu = act;
            u.bn_eps = 1e-3f; u.bn_mom = 0.01f; u.bn_idx = add_bn(L, name, cch); u.out = new_act(cch, hh, ww, false);
            U.push_back(u);
            return u.out;
        };
        auto pw_conv = [&](const std::string& name, int src, int cin, int cout, int hh, int ww) {    // 1x1, no bias, no norm of its own
            Unit u; u.kind = U_CONV; u.src0 = src; u.cin0 = cin; u.cout = cout; u.k = 1; u.pad = 0; u.stride = 1; u.hin = u.hout = hh; u.win = u.wout = ww;
            u.relu = 0; u.frozen_candidate = true; u.w_idx = (int)L.tensors.size(); add_tensor(L, name, {cout, cin, 1, 1}, 0);
            u.out = new_act(cout, hh, ww, false);
            U.push_back(u);
            return u.out;
        };
        auto dw_conv = [&](const std::string& name, int src, int cch, int k, int st, int hh, int ww, int bcast, int dil = 1) {
            Unit u; u.kind = U_DWCONV2; u.src0 = src; u.cin0 = bcast ? 1 : cch; u.cout = cch; u.k = k; u.stride = st; u.dil = dil;
            u.pad = dil > 1 ? (k / 2) * dil : (st == 2 ? (k == 3 ? 0 : 1) : k / 2);
            u.hin = hh; u.win = ww; u.hout = hh / st; u.wout = ww / st; u.bcast = bcast; u.relu = 0; u.frozen_candidate = true;
            u.w_idx = (int)L.tensors.size(); add_tensor(L, name, {cch, 1, k, k}, 0);
            u.out = new_act(cch, hh / st, ww / st, false);
            U.push_back(u);
            return u.out;
        };
        const int stem_c = round_filters(32);
        cur = dw_conv("encoder._conv_stem.weight", -1, stem_c, 3, 2, H, W, 1);
        cur = bn_unit("encoder._bn0", cur, stem_c, H / 2, W / 2, 2);
        feat[1] = cur; featc[1] = stem_c;
        ch = H / 2; cw = W / 2;
        int nblocks = 0;
        for (auto& b : base) nblocks += round_repeats(b[0]);
        int bi = 0, stage = 0;
        for (auto& b : base) {
            const int k = b[1], e = b[3], o = round_filters(b[5]);
            for (int j = 0; j < round_repeats(b[0]); ++j, ++bi) {
                // DeepLabV3+ (output stride 16): smp's replace_strides_with_dilation on the last stage (blocks[ends[2]:]) - every
                // convolution stride 1, dilation 2, padding (k / 2) * 2, the static padding dropped ("Kostyl for EfficientNet")
                // DeepLabV3 (output stride 8): stages 4 and 5 (blocks[ends[1]:ends[2]], blocks[ends[2]:]) with dilation 2 and 4
                // PAN (encoder_dilation): the last stage with dilation 2, as DeepLabV3+
                const int stage_dil = (net->topology == 4 || net->topology == 7) ? (bi >= ends[2] ? 2 : 1)
                                      : net->topology == 5 ? (bi >= ends[2] ? 4 : (bi >= ends[1] ? 2 : 1)) : 1;
                const bool dilated = stage_dil > 1;
                const int st = (j == 0 && !dilated) ? b[2] : 1, inp = j == 0 ? round_filters(b[4]) : o, oup = inp * e;
                const std::string pre = "encoder._blocks." + std::to_string(bi);
                const int x_in = cur;
                int t = cur;
                if (e != 1) {
                    t = pw_conv(pre + "._expand_conv.weight", t, inp, oup, ch, cw);
                    t = bn_unit(pre + "._bn0", t, oup, ch, cw, 2);
                }
                t = dw_conv(pre + "._depthwise_conv.weight", t, oup, k, st, ch, cw, 0, stage_dil);
                const int oh = ch / st, ow = cw / st;
                t = bn_unit(pre + "._bn1", t, oup, oh, ow, 2);
                const int R = std::max(1, inp / 4);
                Unit gp; gp.kind = U_GAP; gp.src0 = t; gp.cout = oup; gp.hin = oh; gp.win = ow; gp.hout = 1; gp.wout = 1; gp.relu = 0;
                gp.out = new_act(oup, 1, 1, false);
                U.push_back(gp);
                Unit se; se.kind = U_SE; se.src0 = gp.out; se.cout = oup; se.cin0 = oup; se.cin1 = R; se.hout = 1; se.wout = 1; se.relu = 2;
                se.w_idx = (int)L.tensors.size();
                add_tensor(L, pre + "._se_reduce.weight", {R, oup, 1, 1}, 0); add_tensor(L, pre + "._se_reduce.bias", {R}, 3);
                add_tensor(L, pre + "._se_expand.weight", {oup, R, 1, 1}, 0); add_tensor(L, pre + "._se_expand.bias", {oup}, 3);
                se.out = new_act(oup, 1, 1, false);
                U.push_back(se);
                Unit cg; cg.kind = U_CGATE; cg.src0 = t; cg.src1 = se.out; cg.cout = oup; cg.hout = oh; cg.wout = ow; cg.relu = 0;
                cg.out = new_act(oup, oh, ow, false);
                U.push_back(cg);
                t = pw_conv(pre + "._project_conv.weight", cg.out, oup, o, oh, ow);
                t = bn_unit(pre + "._bn2", t, o, oh, ow, 0);
                if (st == 1 && inp == o) {
                    Unit da; da.kind = U_DROPADD; da.src0 = t; da.src1 = x_in; da.cout = o; da.hin = da.hout = oh; da.win = da.wout = ow; da.relu = 0;
                    da.drop_p = 0.2f * (float)bi / (float)nblocks; da.salt = bi;
                    da.out = new_act(o, oh, ow, false);
                    U.push_back(da);
                    t = da.out;
                }
                cur = t; ch = oh; cw = ow; inpl = o;
                if (stage < 4 && bi + 1 == ends[stage]) { feat[stage + 2] = cur; featc[stage + 2] = o; ++stage; }
            }
        }
        add_tensor(L, "encoder._conv_head.weight", {round_filters(1280), inpl, 1, 1}, 0);     // registered by efficientnet-pytorch, never run
        add_bn(L, "encoder._bn1", round_filters(1280));
    } else if (net->encoder == 150 || net->encoder == 201) {
        // smp's timm-resnest50d / timm-resnest101e (timm 0.4.12 ResNet(ResNestBottleneck, stem_type 'deep', avg_down, radix 2, avd)):
        // deep stem conv1 = [3x3 / 2 (1 -> sw) BN ReLU, 3x3 (sw -> sw) BN ReLU, 3x3 (sw -> 2 sw)], bn1, ReLU, MaxPool(3, 2, 1); blocks = conv1 1x1 +
        // bn1 + ReLU, conv2 = SplitAttnConv2d (3x3 onto 2 C channels in two groups + bn0 + ReLU; the splits summed and average-pooled;
        // fc1 (bias) + bn1 + ReLU; fc2 (bias); RadixSoftmax; the attention-weighted sum of the splits), avd_last = AvgPool2d(3, 2, 1) in
        // the stride-2 blocks, conv3 1x1 + bn3 (+ shortcut, ReLU); shortcut = [AvgPool2d(2, 2)] + 1x1 + BN.  The reference's freeze
        // predicate ("encoder" and "conv" in the name) also takes conv2.bn0 / conv2.fc1 / conv2.bn1 / conv2.fc2 and the stem's BatchNorms.
        const bool e101 = net->encoder == 201;
        const int sw = e101 ? 64 : 32;
        const int blocks_n[4] = {3, 4, e101 ? 23 : 6, 3};
        auto conv_bn = [&](const std::string& wname, const std::string& bnname, int src, int cin, int cout, int k, int hh, int ww, int relu,
                           bool frozen, bool bn_frozen, int two_groups = 0) {
            Unit u; u.kind = U_CONV; u.src0 = src; u.cin0 = cin; u.cout = cout; u.k = k; u.pad = k / 2; u.stride = 1; u.g2 = two_groups;
            u.hin = hh; u.win = ww; u.hout = hh; u.wout = ww; u.relu = relu; u.frozen_candidate = frozen; u.aux_frozen = bn_frozen;
            u.w_idx = (int)L.tensors.size(); add_tensor(L, wname, {cout, two_groups ? cin / 2 : cin, k, k}, 0);
            u.bn_idx = add_bn(L, bnname, cout);
            u.out = new_act(cout, hh, ww, true);
            return u;
        };
        {   // deep stem
            Unit d; d.kind = U_DWCONV2; d.src0 = -1; d.cin0 = 1; d.cout = sw; d.k = 3; d.stride = 2; d.pad = 1; d.bcast = 1; d.relu = 0; d.frozen_candidate = true;
            d.hin = H; d.win = W; d.hout = H / 2; d.wout = W / 2;
            d.w_idx = (int)L.tensors.size(); add_tensor(L, "encoder.conv1.0.weight", {sw, 1, 3, 3}, 0);
            d.out = new_act(sw, H / 2, W / 2, false);
            U.push_back(d);
            Unit b; b.kind = U_BN; b.src0 = d.out; b.cin0 = sw; b.cout = sw; b.hin = b.hout = H / 2; b.win = b.wout = W / 2; b.relu = 1; b.aux_frozen = true;
            b.bn_idx = add_bn(L, "encoder.conv1.1", sw); b.out = new_act(sw, H / 2, W / 2, false);
            U.push_back(b);
            Unit c3 = conv_bn("encoder.conv1.3.weight", "encoder.conv1.4", b.out, sw, sw, 3, H / 2, W / 2, 1, true, true);
            U.push_back(c3);
            Unit c6 = conv_bn("encoder.conv1.6.weight", "encoder.bn1", c3.out, sw, 2 * sw, 3, H / 2, W / 2, 1, true, false);
            U.push_back(c6);
            feat[1] = c6.out; featc[1] = 2 * sw;
            Unit pool; pool.kind = U_POOL; pool.src0 = c6.out; pool.cout = 2 * sw; pool.hin = H / 2; pool.win = W / 2;
            pool.hout = H / 4; pool.wout = W / 4; pool.out = new_act(2 * sw, H / 4, W / 4, false);
            U.push_back(pool);
            cur = pool.out; inpl = 2 * sw; ch = H / 4; cw = W / 4;
        }
        const int planes_r[4] = {64, 128, 256, 512};
        for (int l = 0; l < 4; ++l) {
            for (int bidx = 0; bidx < blocks_n[l]; ++bidx) {
                const std::string pre = "encoder.layer" + std::to_string(l + 1) + "." + std::to_string(bidx);
                const int stride = (bidx == 0 && l > 0) ? 2 : 1, C = planes_r[l], outc = 4 * C, A = std::max(C * 2 / 4, 32);
                const int oh = ch / stride, ow = cw / stride;
                const int x_in = cur;
                Unit u1 = conv_bn(pre + ".conv1.weight", pre + ".bn1", cur, inpl, C, 1, ch, cw, 1, true, false);
                U.push_back(u1);
                Unit u2 = conv_bn(pre + ".conv2.conv.weight", pre + ".conv2.bn0", u1.out, C, 2 * C, 3, ch, cw, 1, true, true, 1);   // [2 C][C / 2][3][3]
                U.push_back(u2);
                Unit gp; gp.kind = U_GAP; gp.src0 = u2.out; gp.cout = 2 * C; gp.hin = ch; gp.win = cw; gp.hout = 1; gp.wout = 1; gp.relu = 0;
                gp.out = new_act(2 * C, 1, 1, false);
                U.push_back(gp);
                Unit fd; fd.kind = U_FOLD2; fd.src0 = gp.out; fd.cout = C; fd.hin = fd.hout = 1; fd.win = fd.wout = 1; fd.relu = 0;
                fd.out = new_act(C, 1, 1, false);
                U.push_back(fd);
                Unit f1; f1.kind = U_CONV; f1.src0 = fd.out; f1.cin0 = C; f1.cout = A; f1.k = 1; f1.pad = 0; f1.hin = f1.hout = 1; f1.win = f1.wout = 1; f1.relu = 1;
                f1.frozen_candidate = true; f1.aux_frozen = true;
                f1.w_idx = (int)L.tensors.size(); add_tensor(L, pre + ".conv2.fc1.weight", {A, C, 1, 1}, 0);
                f1.bias_idx = (int)L.tensors.size(); add_tensor(L, pre + ".conv2.fc1.bias", {A}, 3);
                f1.bn_idx = add_bn(L, pre + ".conv2.bn1", A);
                f1.out = new_act(A, 1, 1, true);
                U.push_back(f1);
                Unit f2; f2.kind = U_CONV; f2.src0 = f1.out; f2.cin0 = A; f2.cout = 2 * C; f2.k = 1; f2.pad = 0; f2.hin = f2.hout = 1; f2.win = f2.wout = 1; f2.relu = 0;
                f2.frozen_candidate = true; f2.aux_frozen = true;
                f2.w_idx = (int)L.tensors.size(); add_tensor(L, pre + ".conv2.fc2.weight", {2 * C, A, 1, 1}, 0);
                f2.bias_idx = (int)L.tensors.size(); add_tensor(L, pre + ".conv2.fc2.bias", {2 * C}, 3);
                f2.out = new_act(2 * C, 1, 1, false);
                U.push_back(f2);
                Unit rs; rs.kind = U_RSOFTMAX; rs.src0 = f2.out; rs.cout = 2 * C; rs.hin = rs.hout = 1; rs.win = rs.wout = 1; rs.relu = 0;
                rs.out = new_act(2 * C, 1, 1, false);
                U.push_back(rs);
                Unit fo; fo.kind = U_RADIXSUM; fo.src0 = u2.out; fo.src1 = rs.out; fo.cout = C; fo.hin = fo.hout = ch; fo.win = fo.wout = cw; fo.relu = 0;
                fo.out = new_act(C, ch, cw, false);
                U.push_back(fo);
                int t = fo.out;
                if (stride == 2) {      // avd_last = nn.AvgPool2d(3, 2, padding=1): the padding zeros count (count_include_pad)
                    Unit ap; ap.kind = U_AVGPOOL; ap.src0 = t; ap.cout = C; ap.k = 3; ap.stride = 2; ap.pad = 1; ap.pool_w = 1.f / 9.f; ap.relu = 0;
                    ap.hin = ch; ap.win = cw; ap.hout = oh; ap.wout = ow; ap.out = new_act(C, oh, ow, false);
                    U.push_back(ap);
                    t = ap.out;
                }
                Unit u3 = conv_bn(pre + ".conv3.weight", pre + ".bn3", t, C, outc, 1, oh, ow, 1, true, false);
                if (bidx == 0) {        // downsample_avg: [AvgPool2d(2, 2, ceil_mode, count_include_pad=False)] + 1x1 convolution + BatchNorm
                    int ds = x_in;
                    if (stride == 2) {
                        Unit ap; ap.kind = U_AVGPOOL; ap.src0 = x_in; ap.cout = inpl; ap.k = 2; ap.stride = 2; ap.pad = 0; ap.pool_w = 0.25f; ap.relu = 0;
                        ap.hin = ch; ap.win = cw; ap.hout = oh; ap.wout = ow; ap.out = new_act(inpl, oh, ow, false);
                        U.push_back(ap);
                        ds = ap.out;
                    }
                    Unit ud = conv_bn(pre + ".downsample.1.weight", pre + ".downsample.2", ds, inpl, outc, 1, oh, ow, 0, false, false);
                    // state-dict order: conv3, bn3, then downsample.* - the tensors of u3 were registered first (conv_bn above)
                    U.push_back(ud);
                    u3.res = ud.out;
                } else {
                    u3.res = x_in;
                }
                U.push_back(u3);
                cur = u3.out; inpl = outc; ch = oh; cw = ow;
            }
            feat[l + 2] = cur; featc[l + 2] = inpl;
        }
    } else {
    add_tensor(L, "encoder.conv1.weight", {64, 1, 7, 7}, 0);
    Unit stem; stem.kind = U_STEM; stem.cout = 64; stem.k = 7; stem.stride = 2; stem.pad = 3;
    stem.hin = H; stem.win = W; stem.hout = H / 2; stem.wout = W / 2;
    stem.w_idx = 0; stem.bn_idx = add_bn(L, "encoder.bn1", 64);
    stem.out = new_act(64, H / 2, W / 2, true); stem.frozen_candidate = true;
    U.push_back(stem);
    feat[1] = stem.out;
    Unit pool; pool.kind = U_POOL; pool.src0 = stem.out; pool.cout = 64; pool.hin = H / 2; pool.win = W / 2;
    pool.hout = H / 4; pool.wout = W / 4; pool.out = new_act(64, H / 4, W / 4, false);
    U.push_back(pool);
    cur = pool.out; inpl = 64; ch = H / 4; cw = W / 4;
    const int planes[4] = {64, 128, 256, 512};
    const int blocks18[4] = {2, 2, 2, 2}, blocks34[4] = {3, 4, 6, 3};   // resnet50 uses the resnet34 block counts
    const int* blocks = net->encoder == 18 ? blocks18 : blocks34;
    const bool bottleneck = net->encoder == 50 || net->encoder == 51;   // 51 = resnext50_32x4d: Bottleneck with groups = 32,
    const int groups = net->encoder == 51 ? 32 : 1;                     // width_per_group = 4 (torchvision): width = planes * 2
    const int expansion = bottleneck ? 4 : 1;
    for (int l = 0; l < 4; ++l) {
        for (int b = 0; b < blocks[l]; ++b) {
            const std::string pre = "encoder.layer" + std::to_string(l + 1) + "." + std::to_string(b);
            // DeepLabV3+ (output stride 16): smp's replace_strides_with_dilation turns layer4's stride into dilation 2 - every
            // convolution of the stage gets stride 1, and the 3x3 ones dilation 2 / padding 2
            // DeepLabV3 (output stride 8): layer3 with dilation 2, layer4 with dilation 4
            // PAN (encoder_dilation=True): layer4 with dilation 2, as DeepLabV3+
            const int stage_dil = (net->topology == 4 || net->topology == 7) ? (l == 3 ? 2 : 1) : (net->topology == 5 ? (l == 2 ? 2 : (l == 3 ? 4 : 1)) : 1);
            const bool dilated = stage_dil > 1;
            const int stride = (b == 0 && l > 0 && !dilated) ? 2 : 1;
            const int oh = ch / stride, ow = cw / stride, pl = planes[l], outc = pl * expansion;
            auto conv_unit = [&](const std::string& name, const std::string& bn, int src, int cin, int cout, int k, int st, int hi, int wi,
                                 bool frozen, int cg) {
                Unit u; u.kind = U_CONV; u.src0 = src; u.cin0 = cin; u.cout = cout; u.k = k; u.pad = k / 2; u.stride = st;
                u.hin = hi; u.win = wi; u.hout = hi / st; u.wout = wi / st; u.frozen_candidate = frozen; u.cg = cg;
                if (dilated && k == 3) { u.dil = stage_dil; u.pad = stage_dil; }
                u.w_idx = (int)L.tensors.size(); add_tensor(L, name, {cout, cg ? cg : cin, k, k}, 0);
                u.bn_idx = add_bn(L, bn, cout);
                u.out = new_act(cout, hi / st, wi / st, true);
                return u;
            };
            // the block's convolutions in torchvision's registration order (= state_dict order): conv1 bn1 conv2 bn2 [conv3 bn3]
            // [downsample.0 downsample.1].  BasicBlock: 3x3 (stride) - 3x3; Bottleneck (v1.5): 1x1 - 3x3 (stride) - 1x1 (x4).
            std::vector<Unit> us;
            if (!bottleneck) {
                us.push_back(conv_unit(pre + ".conv1.weight", pre + ".bn1", cur, inpl, pl, 3, stride, ch, cw, true, 0));
                us.push_back(conv_unit(pre + ".conv2.weight", pre + ".bn2", us[0].out, pl, pl, 3, 1, oh, ow, true, 0));
            } else {
                const int width = groups > 1 ? pl * 2 : pl;
                us.push_back(conv_unit(pre + ".conv1.weight", pre + ".bn1", cur, inpl, width, 1, 1, ch, cw, true, 0));
                us.push_back(conv_unit(pre + ".conv2.weight", pre + ".bn2", us[0].out, width, width, 3, stride, ch, cw, true,
                                       groups > 1 ? width / groups : 0));
                us.push_back(conv_unit(pre + ".conv3.weight", pre + ".bn3", us[1].out, width, outc, 1, 1, oh, ow, true, 0));
            }
            Unit& last = us.back();
            for (size_t q = 0; q + 1 < us.size(); ++q) U.push_back(us[q]);
            if (stride != 1 || inpl != outc || (dilated && b == 0)) {   // "downsample" lacks "conv" in its name: not frozen by the reference's predicate
                Unit ud = conv_unit(pre + ".downsample.0.weight", pre + ".downsample.1", cur, inpl, outc, 1, stride, ch, cw, false, 0);
                ud.relu = 0;
                U.push_back(ud);
                last.res = ud.out;
            } else {
                last.res = cur;
            }
            U.push_back(last);
            cur = last.out; inpl = outc; ch = oh; cw = ow;
        }
        feat[l + 2] = cur;
        featc[l + 2] = inpl;
    }
    }
    // ---- decoder ----
    const int dec[5] = {256, 128, 64, 32, 16};
    const int skipc[5] = {featc[4], featc[3], featc[2], featc[1], 0};
    int xin = feat[5], xc = featc[5];
    if (net->topology >= 2) {
        // (built below, next to the head)
    } else if (net->topology == 0) {
    for (int i = 0; i < 5; ++i) {
        const std::string pre = "decoder.blocks." + std::to_string(i);
        const int oh = ch * 2, ow = cw * 2;
        Unit u1; u1.kind = U_CONV; u1.src0 = xin; u1.up0 = 1; u1.cin0 = xc; u1.cin1 = skipc[i];
        u1.src1 = skipc[i] ? feat[4 - i] : -1; u1.cout = dec[i];
        u1.hin = oh; u1.win = ow; u1.hout = oh; u1.wout = ow;
        u1.w_idx = (int)L.tensors.size(); add_tensor(L, pre + ".conv1.0.weight", {dec[i], xc + skipc[i], 3, 3}, 0);
        u1.bn_idx = add_bn(L, pre + ".conv1.1", dec[i]);
        u1.out = new_act(dec[i], oh, ow, true);
        Unit u2; u2.kind = U_CONV; u2.src0 = u1.out; u2.cin0 = dec[i]; u2.cout = dec[i];
        u2.hin = oh; u2.win = ow; u2.hout = oh; u2.wout = ow;
        u2.w_idx = (int)L.tensors.size(); add_tensor(L, pre + ".conv2.0.weight", {dec[i], dec[i], 3, 3}, 0);
        u2.bn_idx = add_bn(L, pre + ".conv2.1", dec[i]);
        u2.out = new_act(dec[i], oh, ow, true);
        U.push_back(u1); U.push_back(u2);
        xin = u2.out; xc = dec[i]; ch = oh; cw = ow;
    }
    } else {
        // smp.UnetPlusPlusDecoder (decoders/unetplusplus/decoder.py of segmentation-models-pytorch 0.2.1, restated):
        // features reversed, deepest first: f[0] = layer4 .. f[4] = stem; in_channels = [C(f0), 256, 128, 64, 32],
        // skip_channels = [C(f1), C(f2), C(f3), C(f4), 0], out_channels = dec.  Node x_d_l = DecoderBlock(up(x_d_(l-1) or f[d]),
        // cat(x_(d+1)_l .. x_l_l, f[l+1])); all nodes x_*_l and f[l+1] share a resolution.  Parameters are REGISTERED in the
        // constructor's order (x_0_0; x_0_1 x_1_1; x_0_2 x_1_2 x_2_2; ..; x_0_4) but EXECUTED in forward()'s order
        // (x_0_0 x_1_1 x_2_2 x_3_3; x_0_1 x_1_2 x_2_3; x_0_2 x_1_3; x_0_3; x_0_4): tensors first, units second.
        const int f_act[5] = {feat[5], feat[4], feat[3], feat[2], feat[1]};
        const int f_c[5] = {featc[5], featc[4], featc[3], featc[2], featc[1]};
        const int in_c[5] = {f_c[0], dec[0], dec[1], dec[2], dec[3]};
        struct Node { int in_ch, skip_ch, out_ch, w1, bn1, w2, bn2, out_act; };
        Node node[5][5] = {};
        auto reg = [&](int d, int l, int in_ch, int skip_ch, int out_ch) {
            const std::string pre = "decoder.blocks.x_" + std::to_string(d) + "_" + std::to_string(l);
            Node& nd = node[d][l];
            nd.in_ch = in_ch; nd.skip_ch = skip_ch; nd.out_ch = out_ch;
            nd.w1 = (int)L.tensors.size(); add_tensor(L, pre + ".conv1.0.weight", {out_ch, in_ch + skip_ch, 3, 3}, 0);
            nd.bn1 = add_bn(L, pre + ".conv1.1", out_ch);
            nd.w2 = (int)L.tensors.size(); add_tensor(L, pre + ".conv2.0.weight", {out_ch, out_ch, 3, 3}, 0);
            nd.bn2 = add_bn(L, pre + ".conv2.1", out_ch);
        };
        for (int l = 0; l < 4; ++l)
            for (int d = 0; d <= l; ++d) {
                if (d == 0) reg(0, l, in_c[l], skipc[l] * (l + 1), dec[l]);
                else reg(d, l, skipc[l - 1], skipc[l] * (l + 1 - d), skipc[l]);
            }
        reg(0, 4, in_c[4], 0, dec[4]);
        auto run_node = [&](int d, int l, int x_act, int x_c, const std::vector<int>& skip_members) {
            Node& nd = node[d][l];
            const Act xa = A[x_act];
            const int oh = xa.h * 2, ow = xa.w * 2;
            if (!skip_members.empty() && x_c % 32 != 0) {   // (EfficientNet features of 136 / 56 / 48 channels as the upsampled input)
                const int cat = materialised_cat(x_act, x_c, skip_members, nd.skip_ch);
                Unit u1; u1.kind = U_CONV; u1.src0 = cat; u1.cin0 = x_c + nd.skip_ch; u1.cout = nd.out_ch; u1.hin = oh; u1.win = ow; u1.hout = oh; u1.wout = ow;
                u1.w_idx = nd.w1; u1.bn_idx = nd.bn1; u1.out = new_act(nd.out_ch, oh, ow, true);
                Unit u2; u2.kind = U_CONV; u2.src0 = u1.out; u2.cin0 = nd.out_ch; u2.cout = nd.out_ch;
                u2.hin = oh; u2.win = ow; u2.hout = oh; u2.wout = ow; u2.w_idx = nd.w2; u2.bn_idx = nd.bn2;
                u2.out = new_act(nd.out_ch, oh, ow, true);
                U.push_back(u1); U.push_back(u2);
                nd.out_act = u2.out;
                return;
            }
            int skip_act = -1;
            if (!skip_members.empty()) {   // the concatenation is materialised (channel-slice copies): its gradient is split back
                Unit cu; cu.kind = U_CONCAT; cu.members = skip_members; cu.cout = nd.skip_ch; cu.hout = oh; cu.wout = ow; cu.relu = 0;
                cu.out = new_act(nd.skip_ch, oh, ow, false);
                U.push_back(cu);
                skip_act = cu.out;
            }
            Unit u1; u1.kind = U_CONV; u1.src0 = x_act; u1.up0 = 1; u1.cin0 = x_c; u1.cin1 = nd.skip_ch; u1.src1 = skip_act;
            u1.cout = nd.out_ch; u1.hin = oh; u1.win = ow; u1.hout = oh; u1.wout = ow; u1.w_idx = nd.w1; u1.bn_idx = nd.bn1;
            u1.out = new_act(nd.out_ch, oh, ow, true);
            Unit u2; u2.kind = U_CONV; u2.src0 = u1.out; u2.cin0 = nd.out_ch; u2.cout = nd.out_ch;
            u2.hin = oh; u2.win = ow; u2.hout = oh; u2.wout = ow; u2.w_idx = nd.w2; u2.bn_idx = nd.bn2;
            u2.out = new_act(nd.out_ch, oh, ow, true);
            U.push_back(u1); U.push_back(u2);
            nd.out_act = u2.out;
        };
        for (int layer = 0; layer < 4; ++layer)
            for (int d = 0; d < 4 - layer; ++d) {
                const int l = d + layer;
                if (layer == 0) {
                    run_node(d, d, f_act[d], f_c[d], {f_act[d + 1]});
                } else {
                    std::vector<int> members;
                    for (int idx = d + 1; idx <= l; ++idx) members.push_back(node[idx][l].out_act);
                    members.push_back(f_act[l + 1]);
                    run_node(d, l, node[d][l - 1].out_act, node[d][l - 1].out_ch, members);
                }
            }
        run_node(0, 4, node[0][3].out_act, node[0][3].out_ch, {});
        xin = node[0][4].out_act; xc = dec[4];
    }
    int head_k = 3, head_h = H, head_w = W;
    if (net->topology == 7) {
        // smp.PAN (decoders/pan/decoder.py of segmentation-models-pytorch 0.2.1, restated; decoder_channels 32, encoder_dilation).
        // ConvBnRelu = biased Conv2d + BatchNorm2d (+ ReLU).  FPABlock(C5, 32): branch1 = AdaptiveAvgPool2d(1) + ConvBnRelu 1x1
        // (broadcast back), mid = ConvBnRelu 1x1, the single-channel pyramid (down1 .. conv1, vs_fpa_pyramid_*), out = plane * mid +
        // branch1.  GAUBlock(Ck, 32)(x, y): conv1 = AdaptiveAvgPool2d(1) + ConvBnRelu 1x1 without ReLU + Sigmoid on y, conv2 =
        // ConvBnRelu 3x3 on x; out = bilinear(y -> x's size) + conv2(x) * conv1(y).  gau3 / gau2 / gau1 on the stride-16 / 8 / 4
        // features; head = Conv2d(32, classes, 3, padding 1) + UpsamplingBilinear2d(4).
        const int C5 = featc[5];
        const Act fa = A[feat[5]];
        auto cbr_bias = [&](const std::string& pre, int src, int cin, int cout, int k, int hh, int ww, int relu) {
            Unit u; u.kind = U_CONV; u.src0 = src; u.cin0 = cin; u.cout = cout; u.k = k; u.pad = k / 2; u.relu = relu;
            u.hin = hh; u.win = ww; u.hout = hh; u.wout = ww;
            u.w_idx = (int)L.tensors.size(); add_tensor(L, pre + ".conv.weight", {cout, cin, k, k}, 0);
            u.bias_idx = (int)L.tensors.size(); add_tensor(L, pre + ".conv.bias", {cout}, 3);
            u.bn_idx = add_bn(L, pre + ".bn", cout);
            u.out = new_act(cout, hh, ww, true);
            return u;
        };
        auto gap = [&](int src, int cch, int hh, int ww) {
            Unit gp; gp.kind = U_GAP; gp.src0 = src; gp.cout = cch; gp.hin = hh; gp.win = ww; gp.hout = 1; gp.wout = 1; gp.relu = 0;
            gp.out = new_act(cch, 1, 1, false);
            U.push_back(gp);
            return gp.out;
        };
        // ---- FPA ----
        Unit b1 = cbr_bias("decoder.fpa.branch1.1", -1, C5, 32, 1, 1, 1, 1);
        Unit mid = cbr_bias("decoder.fpa.mid.0", feat[5], C5, 32, 1, fa.h, fa.w, 1);
        Unit fpa; fpa.kind = U_FPA; fpa.src0 = feat[5]; fpa.cin0 = C5; fpa.cout = 32; fpa.hin = fa.h; fpa.win = fa.w; fpa.hout = fa.h; fpa.wout = fa.w; fpa.relu = 0;
        {
            const char* names[6] = {"decoder.fpa.down1.1", "decoder.fpa.down2.1", "decoder.fpa.down3.1", "decoder.fpa.down3.2", "decoder.fpa.conv2", "decoder.fpa.conv1"};
            const int ks[6] = {7, 5, 3, 3, 5, 7};
            for (int i = 0; i < 6; ++i) {
                fpa.tens.push_back((int)L.tensors.size()); add_tensor(L, std::string(names[i]) + ".conv.weight", {1, i == 0 ? C5 : 1, ks[i], ks[i]}, 0);
                fpa.tens.push_back((int)L.tensors.size()); add_tensor(L, std::string(names[i]) + ".conv.bias", {1}, 3);
                const int g = add_bn(L, std::string(names[i]) + ".bn", 1);
                fpa.tens.push_back(g); fpa.tens.push_back(g + 1);
            }
            fpa.w_idx = fpa.tens[0];
        }
        b1.src0 = gap(feat[5], C5, fa.h, fa.w);
        U.push_back(b1); U.push_back(mid);
        fpa.src1 = mid.out; fpa.res = b1.out;
        fpa.out = new_act(32, fa.h, fa.w, false);
        U.push_back(fpa);
        // ---- GAU x 3 ----
        int y_act = fpa.out;
        const int gx[3] = {feat[4], feat[3], feat[2]};
        const int gc[3] = {featc[4], featc[3], featc[2]};
        for (int i = 0; i < 3; ++i) {
            const std::string pre = "decoder.gau" + std::to_string(3 - i);
            const Act xa = A[gx[i]], ya = A[y_act];
            Unit c1 = cbr_bias(pre + ".conv1.1", -1, 32, 32, 1, 1, 1, 0);      // registered first, as in smp's constructor
            Unit c2 = cbr_bias(pre + ".conv2", gx[i], gc[i], 32, 3, xa.h, xa.w, 1);
            c1.src0 = gap(y_act, 32, ya.h, ya.w);
            U.push_back(c1);
            Unit sg; sg.kind = U_SIGMOID; sg.src0 = c1.out; sg.cout = 32; sg.hout = 1; sg.wout = 1; sg.relu = 0;
            sg.out = new_act(32, 1, 1, false);
            U.push_back(sg);
            U.push_back(c2);
            Unit cg; cg.kind = U_CGATE; cg.src0 = c2.out; cg.src1 = sg.out; cg.cout = 32; cg.hout = xa.h; cg.wout = xa.w; cg.relu = 0;
            cg.out = new_act(32, xa.h, xa.w, false);
            U.push_back(cg);
            Unit up; up.kind = U_BILINEAR; up.src0 = y_act; up.cout = 32; up.factor = xa.h / ya.h; up.hin = ya.h; up.win = ya.w; up.hout = xa.h; up.wout = xa.w; up.relu = 0;
            up.out = new_act(32, xa.h, xa.w, false);
            U.push_back(up);
            Unit ad; ad.kind = U_ADD; ad.src0 = up.out; ad.src1 = cg.out; ad.cout = 32; ad.hout = xa.h; ad.wout = xa.w; ad.relu = 0;
            ad.out = new_act(32, xa.h, xa.w, false);
            U.push_back(ad);
            y_act = ad.out;
        }
        xin = y_act; xc = 32; head_k = 3; head_h = A[y_act].h; head_w = A[y_act].w;
        net->head_up = 4;
    }
    if (net->topology == 6) {
        // smp.MAnet (decoders/manet/decoder.py of segmentation-models-pytorch 0.2.1, restated): center = PAB(C5, pab_channels 64):
        // top / center 1x1 convs (C5 -> 64), bottom 3x3 conv (C5 -> C5), all biased, attention (vs_pab_attention_*), out_conv 3x3
        // (biased).  blocks[i] = MFAB(in, skip, out, reduction 16) for the four levels with a skip: hl_conv = Conv3x3(in, in) + BN +
        // ReLU, Conv1x1(in, skip) + BN + ReLU; nearest x2; SE_hl on it, SE_ll on the skip (AdaptiveAvgPool2d(1), Conv1x1(skip,
        // skip / 16), ReLU, Conv1x1(-> skip), Sigmoid); x * (SE_hl + SE_ll); cat skip; conv1, conv2 (3x3 + BN + ReLU);
        // blocks[4] = U-Net's DecoderBlock(32, 0, 16).  The gate commutes with the nearest upsampling, so it is applied at the
        // low resolution and conv1 reads up(x * gate) ++ skip through its loader, as the U-Net decoder does.
        const int C5 = featc[5];
        const Act fa = A[feat[5]];
        auto plain = [&](const std::string& name, int src, int cin, int cout, int k) {
            Unit u; u.kind = U_CONV; u.src0 = src; u.cin0 = cin; u.cout = cout; u.k = k; u.pad = k / 2; u.relu = 0;
            u.hin = fa.h; u.win = fa.w; u.hout = fa.h; u.wout = fa.w;
            u.w_idx = (int)L.tensors.size(); add_tensor(L, name + ".weight", {cout, cin, k, k}, 0);
            u.bias_idx = (int)L.tensors.size(); add_tensor(L, name + ".bias", {cout}, 3);
            u.out = new_act(cout, fa.h, fa.w, false);
            U.push_back(u);
            return u.out;
        };
        const int top = plain("decoder.center.top_conv", feat[5], C5, 64, 1);
        const int cen = plain("decoder.center.center_conv", feat[5], C5, 64, 1);
        const int bot = plain("decoder.center.bottom_conv", feat[5], C5, C5, 3);
        Unit pab; pab.kind = U_PAB; pab.src0 = feat[5]; pab.members = {top, cen, bot}; pab.cout = C5; pab.cin0 = 64; pab.hout = fa.h; pab.wout = fa.w; pab.relu = 0;
        pab.out = new_act(C5, fa.h, fa.w, false);
        U.push_back(pab);
        int x_act = plain("decoder.center.out_conv", pab.out, C5, C5, 3), x_c = C5, xh = fa.h, xw = fa.w;
        const int decm[5] = {256, 128, 64, 32, 16};
        const int skipa[5] = {feat[4], feat[3], feat[2], feat[1], -1};
        const int skipcm[5] = {featc[4], featc[3], featc[2], featc[1], 0};
        auto cbr = [&](const std::string& wname, const std::string& bnname, int src, int cin, int cout, int k, int hh, int ww) {
            Unit u; u.kind = U_CONV; u.src0 = src; u.cin0 = cin; u.cout = cout; u.k = k; u.pad = k / 2;
            u.hin = hh; u.win = ww; u.hout = hh; u.wout = ww;
            u.w_idx = (int)L.tensors.size(); add_tensor(L, wname, {cout, cin, k, k}, 0);
            u.bn_idx = add_bn(L, bnname, cout);
            u.out = new_act(cout, hh, ww, true);
            return u;
        };
        for (int i = 0; i < 5; ++i) {
            const std::string pre = "decoder.blocks." + std::to_string(i) + ".";
            const int S = skipcm[i], oc = decm[i];
            int up_src = x_act, up_c = x_c;
            if (S > 0) {
                Unit h0 = cbr(pre + "hl_conv.0.0.weight", pre + "hl_conv.0.1", x_act, x_c, x_c, 3, xh, xw);
                U.push_back(h0);
                Unit h1 = cbr(pre + "hl_conv.1.0.weight", pre + "hl_conv.1.1", h0.out, x_c, S, 1, xh, xw);
                U.push_back(h1);
                const int R = std::max(1, S / 16);
                auto se_tensors = [&](const std::string& nm) {
                    const int w = (int)L.tensors.size();
                    add_tensor(L, pre + nm + ".1.weight", {R, S, 1, 1}, 0); add_tensor(L, pre + nm + ".1.bias", {R}, 3);
                    add_tensor(L, pre + nm + ".3.weight", {S, R, 1, 1}, 0); add_tensor(L, pre + nm + ".3.bias", {S}, 3);
                    return w;
                };
                const int w_ll = se_tensors("SE_ll"), w_hl = se_tensors("SE_hl");      // registration order: ll, then hl
                auto gate = [&](int src, int hh, int ww, int widx) {
                    Unit gp; gp.kind = U_GAP; gp.src0 = src; gp.cout = S; gp.hin = hh; gp.win = ww; gp.hout = 1; gp.wout = 1; gp.relu = 0;
                    gp.out = new_act(S, 1, 1, false);
                    U.push_back(gp);
                    Unit se; se.kind = U_SE; se.src0 = gp.out; se.cout = S; se.cin0 = S; se.cin1 = R; se.w_idx = widx; se.hout = 1; se.wout = 1; se.relu = 0;
                    se.out = new_act(S, 1, 1, false);
                    U.push_back(se);
                    return se.out;
                };
                const int a_hl = gate(h1.out, xh, xw, w_hl);
                const int a_ll = gate(skipa[i], 2 * xh, 2 * xw, w_ll);
                Unit ad; ad.kind = U_ADD; ad.src0 = a_hl; ad.src1 = a_ll; ad.cout = S; ad.hout = 1; ad.wout = 1; ad.relu = 0;
                ad.out = new_act(S, 1, 1, false);
                U.push_back(ad);
                Unit cg; cg.kind = U_CGATE; cg.src0 = h1.out; cg.src1 = ad.out; cg.cout = S; cg.hout = xh; cg.wout = xw; cg.relu = 0;
                cg.out = new_act(S, xh, xw, false);
                U.push_back(cg);
                up_src = cg.out; up_c = S;
            }
            xh *= 2; xw *= 2;
            Unit c1 = cbr(pre + "conv1.0.weight", pre + "conv1.1", up_src, up_c + S, oc, 3, xh, xw);
            if (S > 0 && up_c % 32 != 0) {     // (EfficientNet skips of 56 / 48 channels: the concatenation is materialised)
                c1.src0 = materialised_cat(up_src, up_c, {skipa[i]}, S);
            } else {
                c1.up0 = 1; c1.cin0 = up_c; c1.cin1 = S; c1.src1 = S > 0 ? skipa[i] : -1;
            }
            U.push_back(c1);
            Unit c2 = cbr(pre + "conv2.0.weight", pre + "conv2.1", c1.out, oc, oc, 3, xh, xw);
            U.push_back(c2);
            x_act = c2.out; x_c = oc;
        }
        xin = x_act; xc = 16;
    }
    if (net->topology == 5) {
        // smp.DeepLabV3 (decoders/deeplabv3/decoder.py, restated; encoder_output_stride 8): DeepLabV3Decoder = Sequential(ASPP(C5, 256,
        // rates (12, 24, 36)), Conv2d(256, 256, 3, padding=1, bias=False), BatchNorm2d, ReLU); ASPP as in DeepLabV3+ but with DENSE
        // dilated 3x3 branches (ASPPConv); head = Conv2d(256, classes, 1) + UpsamplingBilinear2d(8).
        const int c5 = feat[5], c5c = featc[5], ah = A[feat[5]].h, aw = A[feat[5]].w;
        auto conv_bn = [&](const std::string& wname, const std::string& bnname, int src, int cin, int cout, int hh, int ww, int k, int rate) {
            Unit u; u.kind = U_CONV; u.src0 = src; u.cin0 = cin; u.cout = cout; u.k = k; u.pad = k / 2; u.colr = rate;
            u.hin = hh; u.win = ww; u.hout = hh; u.wout = ww;
            u.w_idx = (int)L.tensors.size(); add_tensor(L, wname, {cout, cin, k, k}, 0);
            u.bn_idx = add_bn(L, bnname, cout);
            u.out = new_act(cout, hh, ww, true);
            U.push_back(u);
            return u.out;
        };
        std::vector<int> branches;
        branches.push_back(conv_bn("decoder.0.convs.0.0.weight", "decoder.0.convs.0.1", c5, c5c, 256, ah, aw, 1, 0));
        const int rates[3] = {12, 24, 36};
        for (int r = 0; r < 3; ++r) {
            const std::string pre = "decoder.0.convs." + std::to_string(r + 1);
            branches.push_back(conv_bn(pre + ".0.weight", pre + ".1", c5, c5c, 256, ah, aw, 3, rates[r]));
        }
        {
            Unit gp; gp.kind = U_GAP; gp.src0 = c5; gp.cout = c5c; gp.hin = ah; gp.win = aw; gp.hout = 1; gp.wout = 1; gp.relu = 0;
            gp.out = new_act(c5c, 1, 1, false);
            U.push_back(gp);
            const int pooled = conv_bn("decoder.0.convs.4.1.weight", "decoder.0.convs.4.2", gp.out, c5c, 256, 1, 1, 1, 0);
            Unit bc; bc.kind = U_BCAST; bc.src0 = pooled; bc.cout = 256; bc.hin = 1; bc.win = 1; bc.hout = ah; bc.wout = aw; bc.relu = 0;
            bc.out = new_act(256, ah, aw, false);
            U.push_back(bc);
            branches.push_back(bc.out);
        }
        Unit cat; cat.kind = U_CONCAT; cat.members = branches; cat.cout = 5 * 256; cat.hout = ah; cat.wout = aw; cat.relu = 0;
        cat.out = new_act(5 * 256, ah, aw, false);
        U.push_back(cat);
        const int proj = conv_bn("decoder.0.project.0.weight", "decoder.0.project.1", cat.out, 5 * 256, 256, ah, aw, 1, 0);
        Unit dr; dr.kind = U_DROPOUT_E; dr.src0 = proj; dr.cout = 256; dr.hout = ah; dr.wout = aw; dr.relu = 0;
        dr.out = new_act(256, ah, aw, false);
        U.push_back(dr);
        const int fused = conv_bn("decoder.1.weight", "decoder.2", dr.out, 256, 256, ah, aw, 3, 0);
        xin = fused; xc = 256; head_k = 1; head_h = ah; head_w = aw;
        net->head_up = 8;
    }
    if (net->topology == 4) {
        // smp.DeepLabV3Plus (decoders/deeplabv3/decoder.py of segmentation-models-pytorch 0.2.1, restated; encoder_output_stride 16:
        // layer4 dilated above).  aspp = Sequential(ASPP(C5, 256, rates (12, 24, 36), separable), SeparableConv2d(256, 256, 3), BN,
        // ReLU); ASPP: convs = [1x1 conv + BN + ReLU, 3 x (SeparableConv2d(C5, 256, 3, dilation r) + BN + ReLU), AdaptiveAvgPool2d(1)
        // + 1x1 conv + BN + ReLU + bilinear back to the map], concat, project = 1x1 conv (1280 -> 256) + BN + ReLU + Dropout(0.5);
        // up = UpsamplingBilinear2d(4); block1 = 1x1 conv (C2 -> 48) + BN + ReLU on the stride-4 feature; concat; block2 =
        // SeparableConv2d(304, 256, 3) + BN + ReLU; head = Conv2d(256, classes, 1) + UpsamplingBilinear2d(4).
        // SeparableConv2d = depthwise 3x3 (dilation = padding) then pointwise 1x1, no norm in between, both without bias.
        const int c5 = feat[5], c5c = featc[5], ah = A[feat[5]].h, aw = A[feat[5]].w;
        auto conv_bn = [&](const std::string& wname, const std::string& bnname, int src, int cin, int cout, int hh, int ww) {
            Unit u; u.kind = U_CONV; u.src0 = src; u.cin0 = cin; u.cout = cout; u.k = 1; u.pad = 0;
            u.hin = hh; u.win = ww; u.hout = hh; u.wout = ww;
            u.w_idx = (int)L.tensors.size(); add_tensor(L, wname, {cout, cin, 1, 1}, 0);
            u.bn_idx = add_bn(L, bnname, cout);
            u.out = new_act(cout, hh, ww, true);
            U.push_back(u);
            return u.out;
        };
        auto separable_bn = [&](const std::string& pre_conv, const std::string& bnname, int src, int cin, int cout, int hh, int ww, int dil) {
            Unit d; d.kind = U_DWCONV; d.src0 = src; d.cin0 = cin; d.cout = cin; d.k = 3; d.dil = dil; d.pad = dil; d.relu = 0;
            d.hin = hh; d.win = ww; d.hout = hh; d.wout = ww;
            d.w_idx = (int)L.tensors.size(); add_tensor(L, pre_conv + ".0.weight", {cin, 1, 3, 3}, 0);
            d.out = new_act(cin, hh, ww, false);
            // the pointwise convolution's weight follows the depthwise one in the state dict, its BatchNorm after both
            Unit u; u.kind = U_CONV; u.src0 = d.out; u.cin0 = cin; u.cout = cout; u.k = 1; u.pad = 0;
            u.hin = hh; u.win = ww; u.hout = hh; u.wout = ww;
            u.w_idx = (int)L.tensors.size(); add_tensor(L, pre_conv + ".1.weight", {cout, cin, 1, 1}, 0);
            u.bn_idx = add_bn(L, bnname, cout);
            u.out = new_act(cout, hh, ww, true);
            U.push_back(d); U.push_back(u);
            return u.out;
        };
        std::vector<int> branches;
        branches.push_back(conv_bn("decoder.aspp.0.convs.0.0.weight", "decoder.aspp.0.convs.0.1", c5, c5c, 256, ah, aw));
        const int rates[3] = {12, 24, 36};
        for (int r = 0; r < 3; ++r) {
            const std::string pre = "decoder.aspp.0.convs." + std::to_string(r + 1);
            branches.push_back(separable_bn(pre + ".0", pre + ".1", c5, c5c, 256, ah, aw, rates[r]));
        }
        {   // ASPPPooling: Sequential(AdaptiveAvgPool2d(1), Conv2d, BatchNorm2d, ReLU) + F.interpolate(size, bilinear, align_corners=False)
            Unit gp; gp.kind = U_GAP; gp.src0 = c5; gp.cout = c5c; gp.hin = ah; gp.win = aw; gp.hout = 1; gp.wout = 1; gp.relu = 0;
            gp.out = new_act(c5c, 1, 1, false);
            U.push_back(gp);
            const int pooled = conv_bn("decoder.aspp.0.convs.4.1.weight", "decoder.aspp.0.convs.4.2", gp.out, c5c, 256, 1, 1);
            Unit bc; bc.kind = U_BCAST; bc.src0 = pooled; bc.cout = 256; bc.hin = 1; bc.win = 1; bc.hout = ah; bc.wout = aw; bc.relu = 0;
            bc.out = new_act(256, ah, aw, false);
            U.push_back(bc);
            branches.push_back(bc.out);
        }
        Unit cat; cat.kind = U_CONCAT; cat.members = branches; cat.cout = 5 * 256; cat.hout = ah; cat.wout = aw; cat.relu = 0;
        cat.out = new_act(5 * 256, ah, aw, false);
        U.push_back(cat);
        const int proj = conv_bn("decoder.aspp.0.project.0.weight", "decoder.aspp.0.project.1", cat.out, 5 * 256, 256, ah, aw);
        Unit dr; dr.kind = U_DROPOUT_E; dr.src0 = proj; dr.cout = 256; dr.hout = ah; dr.wout = aw; dr.relu = 0;
        dr.out = new_act(256, ah, aw, false);
        U.push_back(dr);
        const int aspp = separable_bn("decoder.aspp.1", "decoder.aspp.2", dr.out, 256, 256, ah, aw, 1);
        Unit up; up.kind = U_BILINEAR; up.src0 = aspp; up.cout = 256; up.factor = 4; up.hin = ah; up.win = aw; up.hout = 4 * ah; up.wout = 4 * aw; up.relu = 0;
        up.out = new_act(256, 4 * ah, 4 * aw, false);
        U.push_back(up);
        const Act hr = A[feat[2]];
        const int high = conv_bn("decoder.block1.0.weight", "decoder.block1.1", feat[2], featc[2], 48, hr.h, hr.w);
        Unit cat2; cat2.kind = U_CONCAT; cat2.members = {up.out, high}; cat2.cout = 256 + 48; cat2.hout = hr.h; cat2.wout = hr.w; cat2.relu = 0;
        cat2.out = new_act(256 + 48, hr.h, hr.w, false);
        U.push_back(cat2);
        const int fused = separable_bn("decoder.block2.0", "decoder.block2.1", cat2.out, 256 + 48, 256, hr.h, hr.w, 1);
        xin = fused; xc = 256; head_k = 1; head_h = hr.h; head_w = hr.w;
        net->head_up = 4;
    }
    if (net->topology == 3) {
        // smp.FPN (decoders/fpn/decoder.py of segmentation-models-pytorch 0.2.1, restated): p5 = Conv1x1(c5); p_k = nearest-x2(p_(k+1)) +
        // Conv1x1(c_k) for k = 4, 3, 2 (pyramid_channels 256, biased, no norm); seg_blocks[i] on p5, p4, p3, p2 with 3, 2, 1, 0
        // upsamplings: Conv3x3(256 -> 128, no bias) + GroupNorm(32) + ReLU (+ bilinear x2, align_corners) then (ups - 1) x
        // [Conv3x3(128 -> 128) + GN + ReLU + bilinear x2]; merge = sum; Dropout2d(0.2); head = Conv1x1(128 -> classes) +
        // UpsamplingBilinear2d(4).
        const int pyr = 256, seg = 128;
        auto lateral = [&](const std::string& name, int src, int cin) {
            const Act sa = A[src];
            Unit u; u.kind = U_CONV; u.src0 = src; u.cin0 = cin; u.cout = pyr; u.k = 1; u.pad = 0; u.relu = 0;
            u.hin = sa.h; u.win = sa.w; u.hout = sa.h; u.wout = sa.w;
            u.w_idx = (int)L.tensors.size(); add_tensor(L, name + ".weight", {pyr, cin, 1, 1}, 0);
            u.bias_idx = (int)L.tensors.size(); add_tensor(L, name + ".bias", {pyr}, 3);
            u.out = new_act(pyr, sa.h, sa.w, false);
            U.push_back(u);
            return u.out;
        };
        int pyramid[4];
        pyramid[0] = lateral("decoder.p5", feat[5], featc[5]);
        for (int k = 0; k < 3; ++k) {   // p4, p3, p2
            const int lat = lateral("decoder.p" + std::to_string(4 - k) + ".skip_conv", feat[4 - k], featc[4 - k]);
            const Act la = A[lat];
            Unit u; u.kind = U_UPADD; u.src0 = pyramid[k]; u.src1 = lat; u.cout = pyr; u.hout = la.h; u.wout = la.w; u.relu = 0;
            u.hin = la.h / 2; u.win = la.w / 2;
            u.out = new_act(pyr, la.h, la.w, false);
            U.push_back(u);
            pyramid[k + 1] = u.out;
        }
        int merged = -1;
        for (int i = 0; i < 4; ++i) {
            const int ups = 3 - i;
            int x_act = pyramid[i], x_c = pyr;
            for (int j = 0; j < std::max(1, ups); ++j) {
                const std::string pre = "decoder.seg_blocks." + std::to_string(i) + ".block." + std::to_string(j) + ".block.";
                const Act xa = A[x_act];
                Unit u; u.kind = U_CONV; u.src0 = x_act; u.cin0 = x_c; u.cout = seg; u.k = 3; u.pad = 1;
                u.hin = xa.h; u.win = xa.w; u.hout = xa.h; u.wout = xa.w;
                u.w_idx = (int)L.tensors.size(); add_tensor(L, pre + "0.weight", {seg, x_c, 3, 3}, 0);
                u.gn_idx = (int)L.tensors.size(); u.gn_groups = 32;
                add_tensor(L, pre + "1.weight", {seg}, 1); add_tensor(L, pre + "1.bias", {seg}, 2);
                u.out = new_act(seg, xa.h, xa.w, true);
                U.push_back(u);
                x_act = u.out; x_c = seg;
                if (ups > 0) {
                    Unit b; b.kind = U_BILINEAR; b.src0 = x_act; b.cout = seg; b.hin = xa.h; b.win = xa.w; b.hout = 2 * xa.h; b.wout = 2 * xa.w;
                    b.relu = 0;
                    b.out = new_act(seg, 2 * xa.h, 2 * xa.w, false);
                    U.push_back(b);
                    x_act = b.out;
                }
            }
            if (merged < 0) merged = x_act;
            else {
                const Act ma = A[merged];
                Unit ua; ua.kind = U_ADD; ua.src0 = merged; ua.src1 = x_act; ua.cout = seg; ua.hout = ma.h; ua.wout = ma.w; ua.relu = 0;
                ua.out = new_act(seg, ma.h, ma.w, false);
                U.push_back(ua);
                merged = ua.out;
            }
        }
        {
            const Act ma = A[merged];
            Unit d; d.kind = U_DROPOUT; d.src0 = merged; d.cout = seg; d.hout = ma.h; d.wout = ma.w; d.relu = 0;
            d.out = new_act(seg, ma.h, ma.w, false);
            U.push_back(d);
            xin = d.out; xc = seg; head_k = 1; head_h = ma.h; head_w = ma.w;
            net->head_up = 4;
        }
    }
    if (net->topology == 2) {
        // smp.Linknet (decoders/linknet/decoder.py of segmentation-models-pytorch 0.2.1, restated): channels = reversed encoder
        // features (deepest first) + [32]; block i = Conv2dReLU(in, in/4, 1) -> TransposeX2(in/4, in/4) -> Conv2dReLU(in/4, out, 1),
        // then + skip (the next-shallower encoder feature) for i < 4; head = Conv2d(32, classes, 1).
        const int chans[6] = {featc[5], featc[4], featc[3], featc[2], featc[1], 32};
        const int skips[5] = {feat[4], feat[3], feat[2], feat[1], -1};
        int x_act = feat[5], xh = A[feat[5]].h, xw = A[feat[5]].w;
        for (int i = 0; i < 5; ++i) {
            const std::string pre = "decoder.blocks." + std::to_string(i) + ".block.";
            const int cin = chans[i], mid = cin / 4, cout = chans[i + 1];
            Unit u1; u1.kind = U_CONV; u1.src0 = x_act; u1.cin0 = cin; u1.cout = mid; u1.k = 1; u1.pad = 0;
            u1.hin = xh; u1.win = xw; u1.hout = xh; u1.wout = xw;
            u1.w_idx = (int)L.tensors.size(); add_tensor(L, pre + "0.0.weight", {mid, cin, 1, 1}, 0);
            u1.bn_idx = add_bn(L, pre + "0.1", mid);
            u1.out = new_act(mid, xh, xw, true);
            U.push_back(u1);
            Unit ut; ut.kind = U_CONVT; ut.src0 = u1.out; ut.cin0 = mid; ut.cout = mid; ut.k = 3; ut.pad = 1;
            ut.hin = xh; ut.win = xw; ut.hout = 2 * xh; ut.wout = 2 * xw;
            ut.w_idx = (int)L.tensors.size(); add_tensor(L, pre + "1.0.weight", {mid, mid, 4, 4}, 3);   // torch's [in][out][kh][kw], as is
            ut.bias_idx = (int)L.tensors.size(); add_tensor(L, pre + "1.0.bias", {mid}, 3);
            ut.bn_idx = add_bn(L, pre + "1.1", mid);
            ut.out = new_act(mid, 2 * xh, 2 * xw, true);
            U.push_back(ut);
            xh *= 2; xw *= 2;
            Unit u3; u3.kind = U_CONV; u3.src0 = ut.out; u3.cin0 = mid; u3.cout = cout; u3.k = 1; u3.pad = 0;
            u3.hin = xh; u3.win = xw; u3.hout = xh; u3.wout = xw;
            u3.w_idx = (int)L.tensors.size(); add_tensor(L, pre + "2.0.weight", {cout, mid, 1, 1}, 0);
            u3.bn_idx = add_bn(L, pre + "2.1", cout);
            u3.out = new_act(cout, xh, xw, true);
            U.push_back(u3);
            x_act = u3.out;
            if (skips[i] >= 0) {
                Unit ua; ua.kind = U_ADD; ua.src0 = u3.out; ua.src1 = skips[i]; ua.cout = cout; ua.hout = xh; ua.wout = xw; ua.relu = 0;
                ua.out = new_act(cout, xh, xw, false);
                U.push_back(ua);
                x_act = ua.out;
            }
        }
        xin = x_act; xc = 32; head_k = 1;
    }
    Unit head; head.kind = U_HEAD; head.src0 = xin; head.cin0 = xc; head.cout = net->classes; head.relu = 0;
    head.k = head_k; head.pad = head_k / 2;
    head.hin = head_h; head.win = head_w; head.hout = head_h; head.wout = head_w;
    head.w_idx = (int)L.tensors.size(); add_tensor(L, "segmentation_head.0.weight", {net->classes, xc, head_k, head_k}, 0);
    head.bias_idx = (int)L.tensors.size(); add_tensor(L, "segmentation_head.0.bias", {net->classes}, 3);
    U.push_back(head);
    return VS_OK;
}

size_t plan_workspace(vs_unet* net) {
    const size_t N = (size_t)net->max_batch, esz = net->esz;
    size_t off = 0;
    // (skewing the tensors' offsets against one another - 4 KB, 68 KB, 1 MB steps - was measured: no effect on the step; what looked like a placement effect was
    // the side stream's hardware queue, see acquire_side_streams)
    auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
    // weight copies + BN constants
    size_t ct = 0;
    for (auto& u : net->units) {
        if (u.kind == U_CONV || u.kind == U_HEAD) {
            const size_t taps = (size_t)u.k * u.k, cin = (size_t)u.cin0 + u.cin1;
            const size_t cout_pad = u.kind == U_HEAD ? 16 : (size_t)u.cout;
            u.off_wc = take((size_t)u.cout * taps * (u.cg ? 32 : cin) * esz);
            u.off_wt = take(cin * taps * (u.cg ? 32 : cout_pad) * esz);
        }
        if (u.kind == U_CONVT) {   // the equivalent 3x3 convolution onto 4 * cout channels
            u.off_wc = take((size_t)4 * u.cout * 9 * u.cin0 * esz);
            u.off_wt = take((size_t)u.cin0 * 9 * 4 * u.cout * esz);
            ct = std::max(ct, N * u.hin * u.win * 4 * u.cout * esz);
        }
        if (u.bn_idx >= 0) u.off_bn = take(4 * (size_t)u.cout * sizeof(float));
    }
    net->off_ct = take(ct);    // the transposed convolutions' un-shuffled output
    for (auto& u : net->units) {
        if (u.kind != U_FPA) continue;
        u.off_fpa_pool = take(2 * N * (u.hin / 2) * (u.win / 2) * u.cin0 * esz);     // the pooled input and (backward) its gradient
        u.off_fpa_arena = take(vs_fpa_arena_floats((int)N, u.hin, u.win) * sizeof(float));
        u.off_fpa_plane = take(2 * N * u.hin * u.win * sizeof(float));     // the plane and (backward) its gradient
    }
    {
        size_t pab = 0, sews = 0, gapws = 0;
        for (auto& u : net->units) {
            if (u.kind == U_PAB) {
                const size_t hw = (size_t)u.hout * u.wout;
                u.off_gn = take(N * hw * hw * sizeof(float));                       // the attention map, kept for the backward pass
                pab = std::max(pab, vs_pab_scratch_bytes((int)N, (int)hw, u.cout));
            }
            if (u.kind == U_SE) u.off_gn = take(N * (size_t)u.cin1 * sizeof(float));   // hidden activations
            if (u.kind == U_SE) sews = std::max(sews, vs_se_gate_scratch_floats((int)N, u.cout, u.cin1) * sizeof(float));
            if (u.kind == U_GAP || u.kind == U_CGATE) gapws = std::max(gapws, vs_sample_rowsum_workspace((int)N, u.cout));
            if (u.kind == U_RADIXSUM) gapws = std::max(gapws, vs_sample_rowsum_workspace((int)N, 2 * u.cout));
            if (u.kind == U_DROPADD) u.off_gn = take(N * sizeof(float));               // the drop-connect draw per sample
        }
        net->pab_bytes = pab;
        net->off_pab = take(pab);
        net->off_sews = take(sews);
        for (auto& u : net->units)
            if (u.kind == U_AVGPOOL) net->avgw_c = std::max(net->avgw_c, u.cout);
        net->off_avgw9 = take((size_t)net->avgw_c * 9 * sizeof(float));
        net->off_avgw4 = take((size_t)net->avgw_c * 4 * sizeof(float));
        net->gapws_bytes = gapws;
        net->off_gapws = take(gapws);
    }
    {
        size_t ys = 0;
        for (auto& u : net->units) {
            if (u.kind != U_CONV || !u.colr) continue;
            const size_t col = N * u.hin * u.win * 9 * u.cin0 * esz;
            u.off_xs = take(col);
            ys = std::max(ys, col);
        }
        net->off_ys = take(ys);
    }
    {   // smp.FPN: GroupNorm statistics per unit, a scratch for the pre-norm convolution output in evaluation (training keeps z),
        // the reduction workspace, the quarter-resolution logits in front of the head's bilinear upsampling
        size_t gnz = 0, gnws = 0;
        for (auto& u : net->units) {
            if (u.kind != U_CONV || u.gn_idx < 0) continue;
            u.off_gn = take((size_t)2 * N * u.gn_groups * sizeof(float));
            gnz = std::max(gnz, N * u.hout * u.wout * u.cout * esz);
            gnws = std::max(gnws, vs_gn_bwd_workspace((int)N, u.cout, u.gn_groups));
        }
        net->off_gnz = take(gnz);
        net->gnws_bytes = gnws;
        net->off_gnws = take(gnws);
        const Unit& hd = net->units.back();
        net->off_lsmall = take(net->head_up > 1 ? N * net->classes * (size_t)hd.hout * hd.wout * sizeof(float) : 0);
    }
    int cmax = 512;
    for (auto& u : net->units) cmax = std::max(cmax, u.cout);
    net->bnws_bytes = 4 * vs_bn_workspace(0, cmax);  // also receives the conv epilogue's per-tile statistics
    net->off_bnws = take(net->bnws_bytes);
    net->off_bncnt = take(256);                      // grid-barrier counters of the one-launch BatchNorm backward (zeroed by vs_unet_prepare)
    net->off_syncsc = take((size_t)2 * cmax * sizeof(float));
    {   // fixed-point statistics bins of every convolution + BatchNorm unit (ConvParams::stats_bins): one contiguous block
        size_t total = 0;
        for (auto& u : net->units)
            if ((u.kind == U_CONV && u.bn_idx >= 0 && u.bias_idx < 0) || (u.kind == U_STEM && u.cout == 64))
                total += unit_bins_bytes(u.cout) + kTicketBytes;
        net->bins_bytes = total;
        net->off_bins0 = take(total);
        size_t at = net->off_bins0;
        for (auto& u : net->units)
            if ((u.kind == U_CONV && u.bn_idx >= 0 && u.bias_idx < 0) || (u.kind == U_STEM && u.cout == 64)) {
                u.off_bins = at; at += unit_bins_bytes(u.cout) + kTicketBytes;
            }
    }
    // activations (a for all, z for conv/stem outputs)
    for (auto& a : net->acts) {
        const size_t bytes = N * a.c * a.h * a.w * esz;
        a.off_a = take(bytes);
    }
    net->off_logits = take(N * net->classes * (size_t)net->h * net->w * sizeof(float));   // vs_unet_forward_to_volume's fallback path
    net->ws_eval = off;
    size_t ctdw = 0;
    for (auto& u : net->units) {
        if (u.kind == U_CONVT) {
            u.off_wc2 = take((size_t)4 * u.cout * 9 * u.cin0 * esz);
            u.off_wt2 = take((size_t)u.cin0 * 9 * 4 * u.cout * esz);
            ctdw = std::max(ctdw, (size_t)4 * u.cout * 9 * u.cin0 * sizeof(float));
        }
        if (u.kind == U_CONV && u.g2) ctdw = std::max(ctdw, (size_t)u.cout * u.k * u.k * u.cin0 * sizeof(float));
        if (u.kind != U_CONV && u.kind != U_HEAD) continue;
        const size_t taps = (size_t)u.k * u.k, cin = (size_t)u.cin0 + u.cin1;
        u.off_wc2 = take((size_t)u.cout * taps * (u.cg ? 32 : cin) * esz);
        u.off_wt2 = take(cin * taps * (u.cg ? 32 : (u.kind == U_HEAD ? 16 : (size_t)u.cout)) * esz);
    }
    {
        const Unit& hd = net->units.back();
        net->off_dlsmall = take(net->head_up > 1 ? N * net->classes * (size_t)hd.hout * hd.wout * sizeof(float) : 0);
        size_t dm = 0;
        for (auto& u : net->units)
            if (u.kind == U_DROPOUT) dm = std::max(dm, N * u.cout * sizeof(float));
        net->off_dropmask = take(dm);
    }
    net->ctdw_bytes = ctdw;
    net->off_ctdw = take(ctdw * vs_unet::kSide);   // dense weight gradient of a transposed convolution's 3x3 form, per side stream
    for (auto& a : net->acts) {
        const size_t bytes = N * a.c * a.h * a.w * esz;
        if (a.has_z) { a.off_z = take(bytes); a.off_dz = take(bytes); }
        a.off_da = take(bytes);
    }
    // wgrad split-K slabs: worst case over layers
    size_t wg = vs_stem_wgrad_workspace((int)N, net->h, net->w);
    for (auto& u : net->units) {
        if (u.kind != U_CONV && u.kind != U_HEAD && u.kind != U_CONVT) continue;
        WgradParams p{};
        p.C0 = u.cin0; p.C1 = u.cin1; p.up0 = u.up0; p.N = (int)N; p.Hin = u.hin; p.Win = u.win;
        p.Hout = u.hout; p.Wout = u.wout; p.stride = u.stride; p.pad = u.pad; p.KH = p.KW = u.k;
        p.Cout = u.kind == U_HEAD ? 16 : u.cout;
        if (u.kind == U_CONVT) { p.Hout = u.hin; p.Wout = u.win; p.Cout = 4 * u.cout; }
        p.cg = u.cg; p.dil = u.dil;
        if (u.colr) { p.C0 = 9 * u.cin0; p.pad = 0; p.KH = p.KW = 1; }
        const size_t b = wgrad_workspace_bytes(net->dtype, p);
        if (b > wg) wg = b;
    }
    for (auto& u : net->units) {
        if (u.kind == U_DWCONV) wg = std::max(wg, vs_dwconv3x3_wgrad_workspace(u.cout));
        if (u.kind == U_DWCONV2) wg = std::max(wg, vs_dwconv2d_wgrad_workspace(u.cout, u.k));
    }
    net->wgws_bytes = wg;
    net->off_wgws = take(wg * vs_unet::kSide);  // one slab workspace per side stream
    {
        const Unit& hd = net->units.back();
        net->off_headdw = take((size_t)16 * hd.k * hd.k * hd.cin0 * sizeof(float));
        net->off_headpart = take((size_t)1024 * 16 * sizeof(float));   // bias-gradient partials of the head's conversion sweep when it runs on the side stream
    }
    net->off_dyh = take(N * net->h * net->w * 16 * esz);
    size_t dup = 0, zs = 0;
    for (auto& u : net->units) {
        if (u.kind != U_CONV) continue;
        if (u.up0) dup = std::max(dup, N * u.hin * u.win * u.cin0 * esz);
        if (u.stride == 2) zs = std::max(zs, N * u.hin * u.win * u.cout * esz);
    }
    net->off_dup = take(dup);
    net->off_zs = take(zs);
    {
        size_t idx = 0;      // the max-pool's argmax bytes (one per output element; 64 channels for the ResNets, 128 for timm-resnest101e's stem)
        for (auto& u : net->units)
            if (u.kind == U_POOL) idx = std::max(idx, N * (size_t)u.hout * u.wout * u.cout);
        net->off_idx = take(idx);
    }
    net->ws_train = off;
    return off;
}

struct Ctx {
    vs_unet* net;
    char* ws;
    const float* params;
    float* bnstate;
    hipStream_t s;
    int n;
    void* a(int id) const { return ws + net->acts[id].off_a; }
    void* z(int id) const { return ws + net->acts[id].off_z; }
    void* da(int id) const { return ws + net->acts[id].off_da; }
    void* dz(int id) const { return ws + net->acts[id].off_dz; }
    const TensorInfo& t(int idx) const { return net->layout.tensors[idx]; }
    const float* P(int idx) const { return params + t(idx).offset; }
    const void* wfwd(const Unit& u) const {  // weights in the compute dtype
        return (net->dtype == VS_F32 && !u.cg && !u.g2) ? (const void*)P(u.w_idx) : (const void*)(ws + wc_off(u, net->wset));
    }
    static size_t wc_off(const Unit& u, int set) { return set ? u.off_wc2 : u.off_wc; }
    static size_t wt_off(const Unit& u, int set) { return set ? u.off_wt2 : u.off_wt; }
    float* bnc(const Unit& u, int which) const { return reinterpret_cast<float*>(ws + u.off_bn) + (size_t)which * u.cout; }
    int64_t rows(const Unit& u) const { return (int64_t)n * u.hout * u.wout; }
};

double conv_flops(const Ctx& c, const Unit& u) {  // algorithmic: 2 * MACs of the (un-padded, un-stuffed) convolution
    return 2.0 * c.n * u.hout * u.wout * (double)u.cout * u.k * u.k * (u.cg ? u.cg : u.cin0 + u.cin1);
}
double act_bytes(const Ctx& c, const Unit& u, int passes) {  // `passes` full sweeps over the unit's output tensor
    return (double)passes * c.n * u.hout * u.wout * u.cout * c.net->esz;
}

ConvParams conv_params(const Ctx& c, const Unit& u) {
    ConvParams p{};
    p.src0 = c.a(u.src0);
    p.src1 = u.src1 >= 0 ? c.a(u.src1) : nullptr;
    p.C0 = u.cin0; p.C1 = u.cin1; p.up0 = u.up0;
    p.N = c.n; p.Hin = u.hin; p.Win = u.win; p.Hout = u.hout; p.Wout = u.wout;
    p.stride = u.stride; p.pad = u.pad; p.KH = p.KW = u.k;
    p.w = c.wfwd(u); p.Cout = u.cout;
    p.gc = u.cg ? 32 : 0;
    p.dil = u.dil;
    if (u.colr) { p.src0 = c.ws + u.off_xs; p.C0 = 9 * u.cin0; p.pad = 0; p.KH = p.KW = 1; }   // the 1x1 form over the column form
    if (u.kind == U_CONVT) {   // its 3x3 form: same-size output, 4 * cout channels, always from the prepared copy
        p.Hout = u.hin; p.Wout = u.win; p.Cout = 4 * u.cout;
        p.w = c.ws + Ctx::wc_off(u, c.net->wset);
    }
    return p;
}

// producer / consumer maps of the activation graph (built once)
void ensure_graph_maps(vs_unet* net) {
    if (!net->producer.empty()) return;
    net->producer.assign(net->acts.size(), -1);
    net->first_consumer.assign(net->acts.size(), 1 << 30);
    net->sole_consumer.assign(net->acts.size(), -1);
    std::vector<int> readers(net->acts.size(), 0);
    for (int k = 0; k < (int)net->units.size(); ++k) {
        const Unit& v = net->units[k];
        if (v.out >= 0) net->producer[v.out] = k;
        auto reads = [&](int a) {
            if (a < 0) return;
            if (k < net->first_consumer[a]) net->first_consumer[a] = k;
            ++readers[a];
            net->sole_consumer[a] = k;
        };
        for (int a : {v.src0, v.src1, v.res}) reads(a);
        for (int a : v.members) reads(a);
    }
    for (size_t a = 0; a < readers.size(); ++a)
        if (readers[a] != 1) net->sole_consumer[a] = -1;
}

// the weight-gradient launch of a convolution unit, without dy / dw / workspace
WgradParams wgrad_params(const Ctx& c, const Unit& u) {
    WgradParams p{};
    p.src0 = c.a(u.src0); p.src1 = u.src1 >= 0 ? c.a(u.src1) : nullptr;
    p.C0 = u.cin0; p.C1 = u.cin1; p.up0 = u.up0; p.N = c.n; p.Hin = u.hin; p.Win = u.win;
    p.Hout = u.hout; p.Wout = u.wout; p.stride = u.stride; p.pad = u.pad; p.KH = p.KW = u.k;
    p.Cout = u.cout;
    p.cg = u.cg; p.dil = u.dil;
    if (u.colr) { p.src0 = c.ws + u.off_xs; p.C0 = 9 * u.cin0; p.pad = 0; p.KH = p.KW = 1; }
    return p;
}

// Normalise-on-load (training, bf16): unit ui is a conv -> BN -> ReLU whose output has exactly ONE reader, a stride-1 3x3
// convolution that takes it as src0 (a BasicBlock's conv1 -> conv2; a decoder block's conv1 -> conv2 -> next block's conv1).
// Then ui runs no normalisation sweep: its statistics stay in their fixed-point bins, the reader's workgroups sum them in their
// prologue, normalise ui's PRE-norm tensor while staging it and leave the normalised activation behind as a by-product for the
// weight gradient (ConvParams::nl_*).  Returns the reader's unit index, or -1.
int nl_consumer(const Ctx& c, int ui) {
    vs_unet* net = c.net;
    const int dt = net->dtype;
    const Unit& u = net->units[ui];
    if (dt != VS_BF16 || !vs_option("nl_fwd") || net->stats_hook) return -1;   // (cross-rank statistics: the sweep finalises)
    if (u.kind != U_CONV || u.relu != 1 || u.res >= 0 || u.colr || u.gn_idx >= 0 || u.bias_idx >= 0 || u.bn_idx < 0 || u.cout > 512 ||
        u.cout > vs_option("nl_max_c")) return -1;    // (the reader's table and per-chunk work grow with the channel count: per-unit table in DESIGN.md)
    ensure_graph_maps(net);
    const int vi = net->sole_consumer[u.out];
    if (vi <= ui) return -1;
    const Unit& v = net->units[vi];
    if (v.kind != U_CONV || v.src0 != u.out || v.src1 == u.out || v.res == u.out || v.colr || v.cg || v.g2) return -1;
    if (!conv_igemm_nl_ok(dt, conv_params(c, v))) return -1;
    return vi;
}

}  // namespace

// ---- parameter table -------------------------------------------------------------------------------
static int with_layout(int classes, int encoder_code, Layout& out) {   // encoder_code = topology * 1000 + encoder
    vs_unet tmp{};
    const int encoder = encoder_code % 1000;
    tmp.classes = classes; tmp.h = 64; tmp.w = 64; tmp.max_batch = 1; tmp.dtype = VS_F32; tmp.esz = 4; tmp.encoder = encoder;
    tmp.topology = encoder_code / 1000;
    VS_REQUIRE(tmp.topology >= 0 && tmp.topology <= 7, "topology must be 0 (U-Net), 1 (U-Net++), 2 (Linknet), 3 (FPN), 4 (DeepLabV3+), 5 (DeepLabV3), 6 (MA-Net) or 7 (PAN), got %d", tmp.topology);
    VS_REQUIRE(classes >= 1 && classes <= 16, "classes must be in [1,16], got %d", classes);
    VS_REQUIRE(encoder == 18 || encoder == 34 || encoder == 50 || encoder == 51 || ((encoder == 103 || encoder == 104) && tmp.topology != 2) ||
               ((encoder == 150 || encoder == 201) && tmp.topology != 4 && tmp.topology != 5 && tmp.topology != 7),
               "encoder must be 18, 34, 50, 51 (resnet18 / resnet34 / resnet50 / resnext50_32x4d), 103 / 104 (efficientnet-b3 / b4; not under Linknet) or "
               "150 / 201 (timm-resnest50d / 101e; not under DeepLabV3(+) / PAN), got %d", encoder_code);
    build(&tmp);
    out = tmp.layout;
    return VS_OK;
}

extern "C" int vs_unet_num_tensors_ex(int classes, int encoder) {
    Layout L;
    if (with_layout(classes, encoder, L)) return VS_ERR_INVALID;
    return (int)L.tensors.size();
}
extern "C" int vs_unet_num_tensors(int classes) { return vs_unet_num_tensors_ex(classes, 34); }

extern "C" int vs_unet_tensor_info_ex(int classes, int encoder, int index, char* name, int name_len, int64_t shape[4], int* ndim,
                                      int* kind, int64_t* offset) {
    Layout L;
    if (with_layout(classes, encoder, L)) return VS_ERR_INVALID;
    VS_REQUIRE(index >= 0 && index < (int)L.tensors.size(), "tensor index %d out of range", index);
    const TensorInfo& t = L.tensors[index];
    if (name && name_len > 0) { strncpy(name, t.name.c_str(), name_len - 1); name[name_len - 1] = 0; }
    for (int i = 0; i < 4; ++i) shape[i] = t.shape[i];
    *ndim = t.ndim; *kind = t.kind; *offset = t.offset;
    return VS_OK;
}

extern "C" int vs_unet_tensor_info(int classes, int index, char* name, int name_len, int64_t shape[4], int* ndim,
                                   int* kind, int64_t* offset) {
    return vs_unet_tensor_info_ex(classes, 34, index, name, name_len, shape, ndim, kind, offset);
}

extern "C" int64_t vs_unet_param_elems_ex(int classes, int encoder) {
    Layout L;
    if (with_layout(classes, encoder, L)) return -1;
    return L.n_params;
}
extern "C" int64_t vs_unet_param_elems(int classes) { return vs_unet_param_elems_ex(classes, 34); }

extern "C" int64_t vs_unet_bnstate_elems_ex(int classes, int encoder) {
    Layout L;
    if (with_layout(classes, encoder, L)) return -1;
    return L.n_bnstate;
}
extern "C" int64_t vs_unet_bnstate_elems(int classes) { return vs_unet_bnstate_elems_ex(classes, 34); }

// ---- lifecycle -------------------------------------------------------------------------------------
extern "C" int vs_unet_create(vs_unet_t** out, int dtype, int classes, int max_batch, int h, int w) {
    return vs_unet_create_ex(out, dtype, classes, max_batch, h, w, 34);
}

extern "C" int vs_unet_create_ex(vs_unet_t** out, int dtype, int classes, int max_batch, int h, int w, int encoder_code) {
    VS_REQUIRE(out, "unet_create: null out pointer");
    const int encoder = encoder_code % 1000, topology = encoder_code / 1000;
    VS_REQUIRE(topology >= 0 && topology <= 7, "unet_create: topology must be 0 (U-Net), 1 (U-Net++), 2 (Linknet), 3 (FPN), 4 (DeepLabV3+), 5 (DeepLabV3), 6 (MA-Net) or 7 (PAN), got %d", topology);
    VS_REQUIRE(encoder == 18 || encoder == 34 || encoder == 50 || encoder == 51 || encoder == 103 || encoder == 104 || encoder == 150 || encoder == 201,
               "unet_create: encoder must be 18, 34, 50, 51, 103, 104, 150 or 201 (resnet18 / resnet34 / resnet50 / resnext50_32x4d / efficientnet-b3 / "
               "efficientnet-b4 / timm-resnest50d / timm-resnest101e), got %d", encoder);
    VS_REQUIRE((encoder != 103 && encoder != 104) || topology != 2,
               "unet_create: the EfficientNet encoders are not built under smp.Linknet (its decoder narrows 56 / 48 channels to 14 / 12: not multiples of 8)");
    VS_REQUIRE((encoder != 150 && encoder != 201) || (topology != 4 && topology != 5 && topology != 7),
               "unet_create: the ResNeSt encoders do not support the dilating decoders (DeepLabV3 / DeepLabV3+ / PAN) - smp's ResNestEncoder.make_dilated "
               "raises for them as well (their parameter-free average pools would stay at stride 2)");
    VS_REQUIRE(dtype == VS_F32 || dtype == VS_BF16 || dtype == VS_F16, "unet_create: bad dtype %d", dtype);
    VS_REQUIRE(classes >= 1 && classes <= 16, "unet_create: classes must be in [1,16], got %d", classes);
    VS_REQUIRE(max_batch >= 1 && h >= 32 && w >= 32 && h % 32 == 0 && w % 32 == 0,
               "unet_create: batch %d, %dx%d - spatial dims must be positive multiples of 32", max_batch, h, w);
    vs_unet* net = new vs_unet();
    net->dtype = dtype; net->classes = classes; net->max_batch = max_batch; net->h = h; net->w = w;
    net->encoder = encoder;
    net->topology = topology;
    net->esz = dtype_size(dtype);
    build(net);
    plan_workspace(net);
    *out = net;
    return VS_OK;
}

extern "C" void vs_unet_destroy(vs_unet_t* net) { delete net; }

extern "C" size_t vs_unet_workspace_bytes(const vs_unet_t* net, int training) {
    return training ? net->ws_train : net->ws_eval;
}

extern "C" int vs_unet_prepare(vs_unet_t* net, const float* params, const float* bnstate, int training, void* workspace,
                               void* stream) {
    VS_REQUIRE(net && params && bnstate && workspace, "unet_prepare: null pointer");
    Ctx c{net, (char*)workspace, params, const_cast<float*>(bnstate), (hipStream_t)stream, 0};
    ProfScope prof(PK_PREPARE, 0, (double)net->layout.n_params * (4 + net->esz * (training ? 2 : 1)), c.s);
    VS_CHECK_HIP(hipMemsetAsync(c.ws + net->off_bncnt, 0, 256, c.s));   // (the counters re-arm themselves; this covers a fresh workspace)
    {   // every conv layer's low-precision copy and flipped/transposed dgrad copy in one launch
        long w_off[64], wc_off[64], wt_off[64];
        int cout[64], taps[64], cin[64], cpad[64], cgs[64], nl = 0;
        auto flush = [&]() -> int {
            if (!nl) return VS_OK;
            const int rc = launch_weight_prepare_all(net->dtype, params, c.ws, nl, w_off, wc_off, wt_off, cout, taps, cin, cpad, cgs, c.s);
            nl = 0;
            return rc;
        };
        for (auto& u : net->units) {
            if (u.kind != U_CONV && u.kind != U_HEAD) continue;
            const bool wc = net->dtype != VS_F32 || u.cg || u.g2, wt = training != 0;
            if (!wc && !wt) continue;
            cgs[nl] = u.g2 ? 255 : u.cg;
            w_off[nl] = c.t(u.w_idx).offset;
            wc_off[nl] = wc ? (long)Ctx::wc_off(u, net->wset) : -1;
            wt_off[nl] = wt ? (long)Ctx::wt_off(u, net->wset) : -1;
            cout[nl] = u.cout; taps[nl] = u.colr ? 1 : u.k * u.k; cin[nl] = u.colr ? 9 * u.cin0 : u.cin0 + u.cin1; cpad[nl] = u.kind == U_HEAD ? 16 : u.cout;
            if (++nl == 64) {   // the descriptor table of one launch holds 64 layers (U-Net++ / resnet50 has 83)
                int rc = flush();
                if (rc) return rc;
            }
        }
        {
            int rc = flush();
            if (rc) return rc;
        }
        for (auto& u : net->units) {
            if (u.kind != U_CONVT) continue;
            const int rc = launch_convt_weight_prepare(net->dtype, c.P(u.w_idx), c.ws + Ctx::wc_off(u, net->wset),
                                                       training ? c.ws + Ctx::wt_off(u, net->wset) : nullptr, u.cin0, u.cout, c.s);
            if (rc) return rc;
        }
    }
    if (net->avgw_c) {     // the average pools' constant taps (the workspace may be a fresh allocation)
        const float w9 = 1.f / 9.f, w4 = 0.25f;
        VS_CHECK_HIP(hipMemsetD32Async((hipDeviceptr_t)(c.ws + net->off_avgw9), *reinterpret_cast<const int*>(&w9), (size_t)net->avgw_c * 9, c.s));
        VS_CHECK_HIP(hipMemsetD32Async((hipDeviceptr_t)(c.ws + net->off_avgw4), *reinterpret_cast<const int*>(&w4), (size_t)net->avgw_c * 4, c.s));
    }
    for (auto& u : net->units) {
        if (u.bn_idx >= 0 && !training) {  // eval-mode folding from the running statistics
            const float* rm = bnstate + c.t(u.bn_idx + 2).offset;
            const float* rv = bnstate + c.t(u.bn_idx + 3).offset;
            int rc = vs_bn_fold(c.P(u.bn_idx), c.P(u.bn_idx + 1), rm, rv, u.kind == U_BN ? u.bn_eps : 1e-5f, c.bnc(u, 0), c.bnc(u, 1), u.cout, stream);
            if (rc) return rc;
            if (u.kind == U_CONV && u.bias_idx >= 0 && (rc = vs_bn_fold_bias(c.bnc(u, 0), c.P(u.bias_idx), c.bnc(u, 1), u.cout, stream))) return rc;
        }
    }
    return VS_OK;
}

// Data-parallel form of the fused optimiser step (the caller all-reduces a bucket of gradients, updates that slice of the
// parameters and then calls this): weight copies of the units [unit_lo, unit_hi) into the plan's OTHER weight set;
// vs_unet_flip_weight_set makes that set current once every range has been refreshed.
extern "C" int vs_unet_prepare_range(vs_unet_t* net, const float* params, void* workspace, void* stream, int unit_lo, int unit_hi) {
    VS_REQUIRE(net && params && workspace && unit_lo >= 0 && unit_lo < unit_hi && unit_hi <= (int)net->units.size(),
               "unet_prepare_range: bad arguments");
    long w_off[64], wc_off[64], wt_off[64];
    int cout[64], taps[64], cin[64], cpad[64], cgs[64], nl = 0;
    const int other = net->wset ^ 1;
    for (int k = unit_lo; k < unit_hi; ++k) {
        const Unit& v = net->units[k];
        if (v.kind == U_CONVT) {
            const int rc = launch_convt_weight_prepare(net->dtype, params + net->layout.tensors[v.w_idx].offset, (char*)workspace + Ctx::wc_off(v, other),
                                                       (char*)workspace + Ctx::wt_off(v, other), v.cin0, v.cout, (hipStream_t)stream);
            if (rc) return rc;
        }
        if (v.kind != U_CONV && v.kind != U_HEAD) continue;
        VS_REQUIRE(nl < 64, "unet_prepare_range: too many layers in one range");
        w_off[nl] = net->layout.tensors[v.w_idx].offset;
        wc_off[nl] = (net->dtype != VS_F32 || v.cg || v.g2) ? (long)Ctx::wc_off(v, other) : -1;
        wt_off[nl] = (long)Ctx::wt_off(v, other);
        cout[nl] = v.cout; taps[nl] = v.colr ? 1 : v.k * v.k; cin[nl] = v.colr ? 9 * v.cin0 : v.cin0 + v.cin1; cpad[nl] = v.kind == U_HEAD ? 16 : v.cout;
        cgs[nl] = v.g2 ? 255 : v.cg;
        ++nl;
    }
    if (!nl) return VS_OK;
    return launch_weight_prepare_all(net->dtype, params, workspace, nl, w_off, wc_off, wt_off, cout, taps, cin, cpad, cgs, (hipStream_t)stream);
}
extern "C" int vs_unet_weight_set(const vs_unet_t* net) { return net ? net->wset : VS_ERR_INVALID; }
// Dropout2d draws of a training forward (smp.FPN): mask = f(seed, *counter) with the counter in device memory (int64; the
// engine passes a BatchNorm's num_batches_tracked, which the training step advances) - a replayed graph draws a new mask per step
extern "C" int vs_unet_set_rng(vs_unet_t* net, uint32_t seed, const int64_t* counter) {
    VS_REQUIRE(net, "unet_set_rng: null pointer");
    net->rng_seed = seed; net->rng_counter = counter;
    return VS_OK;
}
// byte offset inside the training workspace of the last training forward's Dropout2d mask ([n][128] fp32), -1 without dropout (tests)
extern "C" int64_t vs_unet_dropout_mask_offset(const vs_unet_t* net) {
    if (!net) return -1;
    for (auto& u : net->units)
        if (u.kind == U_DROPOUT) return (int64_t)net->off_dropmask;
    return -1;
}
// the drop-connect draws of the last training forward (EfficientNet encoders): for every MBConv block with a skip and a non-zero rate,
// its block index, rate and the byte offset in the training workspace of its [n] fp32 mask (0 or 1 / keep); returns their number (tests)
extern "C" int vs_unet_drop_connect_masks(const vs_unet_t* net, int64_t* offsets, int* blocks, float* rates, int cap) {
    if (!net) return 0;
    int k = 0;
    for (auto& u : net->units) {
        if (u.kind != U_DROPADD || u.drop_p <= 0.f) continue;
        if (k < cap) { if (offsets) offsets[k] = (int64_t)u.off_gn; if (blocks) blocks[k] = u.salt; if (rates) rates[k] = u.drop_p; }
        ++k;
    }
    return k;
}
extern "C" int vs_unet_side_stream(vs_unet_t* net, void* stream, int index, void** side) {
    VS_REQUIRE(net && side && index >= 0 && index < vs_unet::kSide, "unet_side_stream: bad arguments");
    const int rc = acquire_side_streams(net, (hipStream_t)stream);
    if (rc) return rc;
    *side = (void*)net->side[index];
    return VS_OK;
}

extern "C" int vs_unet_side_stream_overlaps(vs_unet_t* net, void* stream, int* overlaps) {
    VS_REQUIRE(net && overlaps, "unet_side_stream_overlaps: null pointer");
    int rc = acquire_side_streams(net, (hipStream_t)stream);
    if (rc) return rc;
    bool yes = false;
    if ((rc = streams_overlap((hipStream_t)stream, net->side[0], &yes))) return rc;
    *overlaps = yes ? 1 : 0;
    return VS_OK;
}

extern "C" int vs_unet_flip_weight_set(vs_unet_t* net) {
    VS_REQUIRE(net, "unet_flip_weight_set: null pointer");
    net->wset ^= 1;
    return VS_OK;
}

// ---- forward ---------------------------------------------------------------------------------------
static int unet_forward(vs_unet_t* net, const float* params, float* bnstate, const float* x, int n, int training,
                        float* logits, void* workspace, void* stream, const VolScatter* scatter);

extern "C" int vs_unet_forward(vs_unet_t* net, const float* params, float* bnstate, const float* x, int n, int training,
                               float* logits, void* workspace, void* stream) {
    VS_REQUIRE(logits, "unet_forward: null pointer");
    return unet_forward(net, params, bnstate, x, n, training, logits, workspace, stream, nullptr);
}

// Prediction: eval-mode forward whose segmentation head writes straight into the output volume(s) - the fused form of
// vs_unet_forward(training = 0) + vs_logits_to_volume (same arithmetic in the same order: identical labels, probabilities
// and keys).  Falls back to exactly those two calls (logits in the plan's workspace) where the fused epilogue does not
// apply: more than 4 classes, vote mode, or slices too small for the direct head kernel.
extern "C" int vs_unet_forward_to_volume(vs_unet_t* net, const float* params, float* bnstate, const float* x, int n,
                                         void* workspace, void* stream, const vs_dirmap* m, int s0, int mode, int direction,
                                         uint8_t* labels, uint16_t* probs, uint32_t* keys, uint8_t* votes, int64_t nvox) {
    VS_REQUIRE(net && m, "unet_forward_to_volume: null pointer");
    VS_REQUIRE(m->hp == net->h && m->wp == net->w, "unet_forward_to_volume: the plan is %dx%d, the padded slices are %dx%d", net->h, net->w, m->hp, m->wp);
    VS_REQUIRE(direction >= 0 && direction < 16, "unet_forward_to_volume: direction %d out of range", direction);
    VolScatter sc{};
    sc.m = *m; sc.s0 = s0; sc.direction = direction; sc.mode = mode; sc.labels = labels; sc.probs = probs; sc.keys = keys;
    float* lg = reinterpret_cast<float*>((char*)workspace + net->off_logits);
    const bool try_fused = mode == 0 || mode == 1;
    // slices along the volume's contiguous axis: the head stages its keys slice-major in the (otherwise unused) logits
    // buffer and a transposing pass merges them into the volume along that axis
    const bool staged = mode == 1 && (m->ss == 1 || m->ss == -1) && m->sw != 1 && m->sw != -1 && n >= 8;
    if (staged) sc.stage = reinterpret_cast<uint32_t*>(lg);
    const int rc = unet_forward(net, params, bnstate, x, n, 0, lg, workspace, stream, try_fused ? &sc : nullptr);
    if (rc < 0) return rc;
    if (try_fused && rc == VS_OK)                 // the head kernel already wrote the volume entries (or staged the keys)
        return staged ? launch_keys_stage_scatter(sc.stage, n, *m, s0, keys, (hipStream_t)stream) : VS_OK;
    return vs_logits_to_volume(lg, net->classes, m, s0, n, mode, direction, labels, probs, keys, votes, nvox, stream);
}

static int unet_forward(vs_unet_t* net, const float* params, float* bnstate, const float* x, int n, int training,
                        float* logits, void* workspace, void* stream, const VolScatter* scatter) {
    VS_REQUIRE(net && params && bnstate && x && logits && workspace, "unet_forward: null pointer");
    VS_REQUIRE(n >= 1 && n <= net->max_batch, "unet_forward: batch %d exceeds the plan's max_batch %d", n, net->max_batch);
    VS_REQUIRE(!(training && net->dtype == VS_F16), "unet_forward: fp16 is the inference precision (batch statistics, the backward pass and the optimiser are built for fp32 / bf16)");
    Ctx c{net, (char*)workspace, params, bnstate, (hipStream_t)stream, n};
    const int dt = net->dtype;
    int rc;
    net->last_n = n;
    bool head_scattered = false;
    int unit_index = -1;
    int carried_stat_rows = 0;     // partial statistic rows a plain convolution's epilogue left in bnws for the U_BN unit right behind it
    bool bn_folded_into_conv = false;   // evaluation: that U_BN's scale / shift / activation already ran in the convolution's epilogue
    net->nl_act.assign(net->acts.size(), 0);
    if (training && dt == VS_BF16 && net->bins_bytes && vs_option("stats_bins"))
        if ((rc = launch_zero_u64((unsigned long long*)(c.ws + net->off_bins0), net->bins_bytes / sizeof(unsigned long long), c.s))) return rc;
    int skip_units = 0;            // evaluation: units that already ran inside the previous unit's launch (conv_pair)
    for (auto& u : net->units) {
        prof_set_tag(++unit_index);
        if (skip_units) { --skip_units; continue; }
        int fused_stat_rows = 0;
        bool fused_bins = false;
        float* rm = u.bn_idx >= 0 ? bnstate + c.t(u.bn_idx + 2).offset : nullptr;
        float* rv = u.bn_idx >= 0 ? bnstate + c.t(u.bn_idx + 3).offset : nullptr;
        switch (u.kind) {
        case U_STEM: {
            ProfScope prof(PK_STEM, 2.0 * n * u.hout * u.wout * 64 * 49, 4.0 * n * net->h * net->w + act_bytes(c, u, 1), c.s);
            if (training && u.cout == 64 && net->bins_bytes && vs_option("stats_bins") && stem_fwd_bins_ok(dt)) {
                // batch statistics from the kernel's own accumulators (fixed-point bins, finalised inside the apply sweep)
                if ((rc = launch_stem_fwd_bins(x, c.P(u.w_idx), c.z(u.out), n, net->h, net->w, (unsigned long long*)(c.ws + u.off_bins),
                                               stat_bins_rows(u.cout), c.s))) return rc;
                fused_bins = true;
            } else if (training && net->stats_hook) {
                vs_set_error("unet_forward: cross-rank BatchNorm statistics need the bf16 stem kernel and the statistics bins (options stem_bf16, stats_bins, fuse_stats)");
                return VS_ERR_UNSUPPORTED;
            } else if (training) {
                if ((rc = vs_stem_fwd(dt, x, c.P(u.w_idx), nullptr, nullptr, 0, c.z(u.out), n, net->h, net->w, stream))) return rc;
            } else {
                if ((rc = vs_stem_fwd(dt, x, c.P(u.w_idx), c.bnc(u, 0), c.bnc(u, 1), 1, c.a(u.out), n, net->h, net->w, stream))) return rc;
            }
            break;
        }
        case U_POOL: {
            ProfScope prof(PK_POOL_MISC, 0, (double)n * u.hin * u.win * 64 * net->esz * 1.25, c.s);
            if ((rc = vs_maxpool_fwd(dt, c.a(u.src0), c.a(u.out), training ? (uint8_t*)(c.ws + net->off_idx) : nullptr, n,
                                     u.hin, u.win, u.cout, stream))) return rc;
            continue;
        }
        case U_CONCAT: {   // torch.cat of U-Net++'s dense skips, materialised (one channel-slice copy per member)
            ProfScope prof(PK_POOL_MISC, 0, 2.0 * n * u.hout * u.wout * u.cout * net->esz, c.s);
            int off = 0;
            for (int m : u.members) {
                const int mc = net->acts[m].c;
                if ((rc = vs_channel_slice(dt, c.a(m), mc, 0, c.a(u.out), u.cout, off, mc, (int64_t)n * u.hout * u.wout, 0, stream))) return rc;
                off += mc;
            }
            continue;
        }
        case U_DWCONV: {
            ProfScope prof(PK_POOL_MISC, 0, 2.0 * n * u.hout * u.wout * u.cout * net->esz, c.s);
            if ((rc = vs_dwconv3x3(dt, c.a(u.src0), c.P(u.w_idx), c.a(u.out), n, u.hin, u.win, u.cout, u.dil, 0, stream))) return rc;
            continue;
        }
        case U_DWCONV2: {
            ProfScope prof(PK_POOL_MISC, 0, (double)n * (u.hin * u.win * u.cin0 + u.hout * u.wout * u.cout) * net->esz, c.s);
            if (!training && !u.bcast && unit_index + 1 < (int)net->units.size() && net->units[unit_index + 1].kind == U_BN &&
                net->units[unit_index + 1].src0 == u.out && (u.dil == 1 || (u.stride == 1 && (u.dil == 2 || (u.dil == 4 && u.k == 3))))) {
                const Unit& bn = net->units[unit_index + 1];      // evaluation: its BatchNorm (folded) + swish in the same sweep
                if ((rc = vs_dwconv2d_affine(dt, c.a(u.src0), c.P(u.w_idx), c.bnc(bn, 0), c.bnc(bn, 1), bn.relu, c.a(bn.out), n, u.hin, u.win, u.cout, u.k,
                                             u.stride, u.pad, u.dil, u.hout, u.wout, stream))) return rc;
                bn_folded_into_conv = true;
                continue;
            }
            if ((rc = vs_dwconv2d(dt, u.bcast ? (const void*)x : (const void*)c.a(u.src0), c.P(u.w_idx), c.a(u.out), n, u.hin, u.win, u.cout, u.k, u.stride,
                                  u.pad, u.dil, u.hout, u.wout, u.bcast, stream))) return rc;
            continue;
        }
        case U_BN: {    // batch statistics (training) or the running ones, normalisation + activation in one sweep
            if (bn_folded_into_conv) { bn_folded_into_conv = false; continue; }
            ProfScope prof(training ? PK_BN_STATS : PK_BN_APPLY, 0, (training ? 3.0 : 2.0) * n * u.hout * u.wout * u.cout * net->esz, c.s);
            const int64_t rows = (int64_t)n * u.hout * u.wout;
            if (training) {
                if (carried_stat_rows) {
                    if ((rc = launch_bn_finalize_partials((const float*)(c.ws + net->off_bnws), carried_stat_rows, u.cout, rows, u.bn_eps, u.bn_mom,
                                                          c.bnc(u, 2), c.bnc(u, 3), rm, rv, c.s))) return rc;
                    carried_stat_rows = 0;
                } else if ((rc = vs_bn2_stats(dt, c.a(u.src0), rows, u.cout, u.bn_eps, u.bn_mom, c.bnc(u, 2), c.bnc(u, 3), rm, rv, (float*)(c.ws + net->off_bnws),
                                       net->bnws_bytes, stream))) return rc;
                if ((rc = vs_bn2_apply(dt, c.a(u.src0), c.bnc(u, 2), c.bnc(u, 3), c.P(u.bn_idx), c.P(u.bn_idx + 1), u.relu, -1.f, c.a(u.out), rows, u.cout, stream))) return rc;
            } else {
                if ((rc = vs_bn2_apply(dt, c.a(u.src0), rm, rv, c.P(u.bn_idx), c.P(u.bn_idx + 1), u.relu, u.bn_eps, c.a(u.out), rows, u.cout, stream))) return rc;
            }
            continue;
        }
        case U_DROPADD: {
            ProfScope prof(PK_POOL_MISC, 0, 3.0 * n * u.hout * u.wout * u.cout * net->esz, c.s);
            const float* mask = nullptr;
            if (training && u.drop_p > 0.f) {
                float* m = (float*)(c.ws + u.off_gn);
                if ((rc = vs_dropout2d_mask(m, n, 1, u.drop_p, net->rng_seed ^ 0x2545f491u, net->rng_counter, (int64_t)u.salt << 32, stream))) return rc;
                mask = m;
            }
            if ((rc = vs_sample_scale_add(dt, c.a(u.src0), mask, c.a(u.src1), c.a(u.out), n, (int64_t)u.hout * u.wout * u.cout, stream))) return rc;
            continue;
        }
        case U_SIGMOID: {
            if ((rc = vs_sigmoid(dt, c.a(u.src0), c.a(u.out), (int64_t)n * u.hout * u.wout * u.cout, stream))) return rc;
            continue;
        }
        case U_FPA: {   // max-pool, 7x7 convolution to the first single-channel map, the pyramid (one workgroup), the combination
            ProfScope prof(PK_POOL_MISC, 0, 3.0 * n * u.hin * u.win * u.cin0 * net->esz, c.s);
            float* arena = (float*)(c.ws + u.off_fpa_arena);
            float* plane = (float*)(c.ws + u.off_fpa_plane);
            if ((rc = vs_maxpool2x2(dt, c.a(u.src0), c.ws + u.off_fpa_pool, n, u.hin, u.win, u.cin0, stream))) return rc;
            if ((rc = vs_conv_to_plane(dt, c.ws + u.off_fpa_pool, c.P(u.tens[0]), c.P(u.tens[1]), arena, n, u.hin / 2, u.win / 2, u.cin0, 7, stream))) return rc;
            float* pp[36];
            for (int i = 0; i < 6; ++i) {
                pp[6 * i + 0] = const_cast<float*>(c.P(u.tens[4 * i + 0])); pp[6 * i + 1] = const_cast<float*>(c.P(u.tens[4 * i + 1]));
                pp[6 * i + 2] = const_cast<float*>(c.P(u.tens[4 * i + 2])); pp[6 * i + 3] = const_cast<float*>(c.P(u.tens[4 * i + 3]));
                pp[6 * i + 4] = bnstate + c.t(u.tens[4 * i + 2] + 2).offset; pp[6 * i + 5] = bnstate + c.t(u.tens[4 * i + 2] + 3).offset;
            }
            if ((rc = vs_fpa_pyramid_fwd(arena, plane, pp, n, u.hin, u.win, training ? 1 : 0, stream))) return rc;
            if ((rc = vs_fpa_combine(dt, plane, c.a(u.src1), c.a(u.res), c.a(u.out), n, (int64_t)u.hout * u.wout, u.cout, stream))) return rc;
            continue;
        }
        case U_PAB: {
            ProfScope prof(PK_POOL_MISC, 0, 4.0 * n * u.hout * u.wout * u.cout * net->esz, c.s);
            if ((rc = vs_pab_attention_fwd(dt, c.a(u.members[0]), c.a(u.members[1]), c.a(u.members[2]), c.a(u.src0), c.a(u.out),
                                           (float*)(c.ws + u.off_gn), (float*)(c.ws + net->off_pab), n, u.hout * u.wout, u.cin0, u.cout, stream))) return rc;
            continue;
        }
        case U_SE: {
            ProfScope prof(PK_POOL_MISC, 4.0 * n * u.cout * u.cin1, 0, c.s);
            if ((rc = vs_se_gate_fwd(dt, c.a(u.src0), c.P(u.w_idx), c.P(u.w_idx + 1), c.P(u.w_idx + 2), c.P(u.w_idx + 3), c.a(u.out),
                                     (float*)(c.ws + u.off_gn), n, u.cout, u.cin1, u.relu == 2, stream))) return rc;
            continue;
        }
        case U_CGATE: {
            ProfScope prof(PK_POOL_MISC, 0, 2.0 * n * u.hout * u.wout * u.cout * net->esz, c.s);
            if ((rc = vs_channel_gate(dt, c.a(u.src0), c.a(u.src1), c.a(u.out), n, (int64_t)u.hout * u.wout, u.cout, stream))) return rc;
            continue;
        }
        case U_GAP: {
            ProfScope prof(PK_POOL_MISC, 0, (double)n * u.hin * u.win * u.cout * net->esz, c.s);
            if ((rc = vs_sample_rowsum_ws(dt, c.a(u.src0), nullptr, c.a(u.out), n, (int64_t)u.hin * u.win, u.cout, 1.f / (float)(u.hin * u.win),
                                          (float*)(c.ws + net->off_gapws), net->gapws_bytes, stream))) return rc;
            continue;
        }
        case U_BCAST: {
            ProfScope prof(PK_POOL_MISC, 0, (double)n * u.hout * u.wout * u.cout * net->esz, c.s);
            if ((rc = vs_broadcast_rows(dt, c.a(u.src0), c.a(u.out), n, (int64_t)u.hout * u.wout, u.cout, 1.f, 0, stream))) return rc;
            continue;
        }
        case U_DROPOUT_E: {
            ProfScope prof(PK_POOL_MISC, 0, 2.0 * n * u.hout * u.wout * u.cout * net->esz, c.s);
            const int64_t elems = (int64_t)n * u.hout * u.wout * u.cout;
            if (training) {
                if ((rc = vs_dropout(dt, c.a(u.src0), c.a(u.out), elems, 0.5f, net->rng_seed ^ 0x5bd1e995u, net->rng_counter, 0, stream))) return rc;
            } else {
                if ((rc = vs_channel_slice(dt, c.a(u.src0), u.cout, 0, c.a(u.out), u.cout, 0, u.cout, (int64_t)n * u.hout * u.wout, 0, stream))) return rc;
            }
            continue;
        }
        case U_FOLD2: {     // the two radix splits summed
            ProfScope prof(PK_POOL_MISC, 0, 3.0 * n * u.hout * u.wout * u.cout * net->esz, c.s);
            const int64_t rows = (int64_t)n * u.hout * u.wout;
            if ((rc = vs_channel_slice(dt, c.a(u.src0), 2 * u.cout, 0, c.a(u.out), u.cout, 0, u.cout, rows, 0, stream))) return rc;
            if ((rc = vs_channel_slice(dt, c.a(u.src0), 2 * u.cout, u.cout, c.a(u.out), u.cout, 0, u.cout, rows, 1, stream))) return rc;
            continue;
        }
        case U_RADIXSUM: {
            ProfScope prof(PK_POOL_MISC, 0, 3.0 * n * u.hout * u.wout * u.cout * net->esz, c.s);
            if ((rc = vs_radix2_gated_sum(dt, c.a(u.src0), c.a(u.src1), c.a(u.out), n, (int64_t)u.hout * u.wout, u.cout, stream))) return rc;
            continue;
        }
        case U_RSOFTMAX: {
            if ((rc = vs_radix2_softmax(dt, c.a(u.src0), c.a(u.out), n, u.cout / 2, stream))) return rc;
            continue;
        }
        case U_AVGPOOL: {
            ProfScope prof(PK_POOL_MISC, 0, (double)n * (u.hin * u.win + u.hout * u.wout) * u.cout * net->esz, c.s);
            const float* taps = (const float*)(c.ws + (u.k == 3 ? net->off_avgw9 : net->off_avgw4));
            if ((rc = vs_dwconv2d(dt, c.a(u.src0), taps, c.a(u.out), n, u.hin, u.win, u.cout, u.k, u.stride, u.pad, 1, u.hout, u.wout, 0, stream))) return rc;
            continue;
        }
        case U_UP2: {
            ProfScope prof(PK_POOL_MISC, 0, 1.25 * n * u.hout * u.wout * u.cout * net->esz, c.s);
            if ((rc = vs_upsample2x_add(dt, c.a(u.src0), nullptr, c.a(u.out), n, u.hin, u.win, u.cout, stream))) return rc;
            continue;
        }
        case U_UPADD: {
            ProfScope prof(PK_POOL_MISC, 0, 2.25 * n * u.hout * u.wout * u.cout * net->esz, c.s);
            if ((rc = vs_upsample2x_add(dt, c.a(u.src0), c.a(u.src1), c.a(u.out), n, u.hin, u.win, u.cout, stream))) return rc;
            continue;
        }
        case U_BILINEAR: {
            ProfScope prof(PK_POOL_MISC, 0, 1.25 * n * u.hout * u.wout * u.cout * net->esz, c.s);
            if ((rc = vs_bilinear_up(dt, c.a(u.src0), c.a(u.out), n, u.hin, u.win, u.cout, u.factor, stream))) return rc;
            continue;
        }
        case U_DROPOUT: {
            ProfScope prof(PK_POOL_MISC, 0, 2.0 * n * u.hout * u.wout * u.cout * net->esz, c.s);
            const int64_t rows = (int64_t)n * u.hout * u.wout;
            if (training) {
                float* mask = (float*)(c.ws + net->off_dropmask);
                if ((rc = vs_dropout2d_mask(mask, n, u.cout, 0.2f, net->rng_seed, net->rng_counter, 0, stream))) return rc;
                if ((rc = vs_channel_scale(dt, c.a(u.src0), mask, c.a(u.out), n, (int64_t)u.hout * u.wout, u.cout, stream))) return rc;
            } else {
                if ((rc = vs_channel_slice(dt, c.a(u.src0), u.cout, 0, c.a(u.out), u.cout, 0, u.cout, rows, 0, stream))) return rc;
            }
            continue;
        }
        case U_CONV: {
            ConvParams p = conv_params(c, u);
            if (training && u.src0 >= 0 && net->nl_act[u.src0]) {   // src0 ran no normalisation sweep: its pre-norm tensor, normalised while it is staged
                const Unit& q = net->units[net->producer[u.src0]];
                p.src0 = c.z(u.src0);
                p.nl_bins = (const unsigned long long*)(c.ws + q.off_bins); p.nl_nb = stat_bins_rows(q.cout);
                p.nl_rows = c.rows(q); p.nl_eps = 1e-5f; p.nl_mom = 0.1f;
                p.nl_mean = c.bnc(q, 2); p.nl_invstd = c.bnc(q, 3);
                p.nl_rm = bnstate + c.t(q.bn_idx + 2).offset; p.nl_rv = bnstate + c.t(q.bn_idx + 3).offset;
                p.nl_gamma = c.P(q.bn_idx); p.nl_beta = c.P(q.bn_idx + 1);
                p.nl_y = c.a(u.src0);      // the activation the weight gradient reads: this launch's by-product
            }
            prof_set_variant(conv_igemm_variant(dt, p));
            ProfScope prof(PK_CONV_FWD, conv_flops(c, u), 0, c.s);
            prof_set_variant(0);
            if (u.bn_idx < 0 && u.gn_idx < 0) {   // plain biased convolution (FPN's lateral 1x1s): no norm, no activation
                p.out = c.a(u.out); p.shift = u.bias_idx >= 0 ? c.P(u.bias_idx) : nullptr;   // (EfficientNet's 1x1 convolutions: no bias either)
                // its BatchNorm is the next unit (U_BN): the batch statistics come straight from the fp32 accumulators, as for the fused units
                if (training && dt == VS_BF16 && u.bias_idx < 0 && unit_index + 1 < (int)net->units.size() &&
                    net->units[unit_index + 1].kind == U_BN && net->units[unit_index + 1].src0 == u.out) {
                    const int rows_needed = conv_igemm_stat_rows(dt, p);
                    if (rows_needed > 0 && (size_t)rows_needed * 2 * u.cout * sizeof(float) <= net->bnws_bytes) {
                        p.stats_partial = (float*)(c.ws + net->off_bnws);
                        carried_stat_rows = rows_needed;
                    }
                }
                // evaluation: the BatchNorm unit behind it (folded running statistics) and its activation ride in this epilogue
                if (!training && u.bias_idx < 0 && unit_index + 1 < (int)net->units.size() && net->units[unit_index + 1].kind == U_BN &&
                    net->units[unit_index + 1].src0 == u.out) {
                    const Unit& bn = net->units[unit_index + 1];
                    p.scale = c.bnc(bn, 0); p.shift = c.bnc(bn, 1); p.relu = bn.relu; p.out = c.a(bn.out);
                    bn_folded_into_conv = true;
                }
                if ((rc = launch_conv_igemm(dt, p, c.s))) return rc;
                continue;
            }
            if (u.colr && (rc = vs_dilated_im2col(dt, c.a(u.src0), c.ws + u.off_xs, n, u.hin, u.win, u.cin0, u.colr, 0, 0, stream))) return rc;
            if (u.gn_idx >= 0) {                  // convolution + GroupNorm + ReLU (per-sample statistics: nothing folds in evaluation)
                p.out = training ? c.z(u.out) : (void*)(c.ws + net->off_gnz);
                if ((rc = launch_conv_igemm(dt, p, c.s))) return rc;
                if ((rc = vs_gn_fwd(dt, p.out, c.P(u.gn_idx), c.P(u.gn_idx + 1), u.relu, c.a(u.out), (float*)(c.ws + u.off_gn), n,
                                    (int64_t)u.hout * u.wout, u.cout, u.gn_groups, 1e-5f, (float*)(c.ws + net->off_gnws), net->gnws_bytes, stream))) return rc;
                continue;
            }
            if (training) {
                p.out = c.z(u.out);
                if (u.bias_idx >= 0) p.shift = c.P(u.bias_idx);   // smp's ConvBnRelu keeps the convolution's bias: z includes it
                if (dt == VS_BF16 && u.bias_idx < 0) {  // batch statistics straight from the fp32 accumulators
                    const int rows_needed = conv_igemm_stat_rows(dt, p);
                    const bool bins_ok = u.bn_idx >= 0 && vs_option("stats_bins") && net->bins_bytes && conv_igemm_bins_ok(dt, p);
                    const bool nl = bins_ok && nl_consumer(c, unit_index) >= 0;
                    if (bins_ok && (nl || net->stats_hook || rows_needed > vs_option("bn_inline_rows"))) {
                        // many tiles: their sums go into a few rows of fixed-point bins, finalised inside the apply sweep -
                        // no finalize launch between the convolution and its normalisation (0.34 ms of a 4.76 ms step)
                        p.stats_bins = (unsigned long long*)(c.ws + u.off_bins);
                        p.stats_nb = stat_bins_rows(u.cout);
                        fused_bins = true;
                        if (nl) net->nl_act[u.out] = 1;   // no sweep: the one reader finalises the bins and normalises on load
                    } else if ((size_t)rows_needed * 2 * u.cout * sizeof(float) <= net->bnws_bytes) {
                        p.stats_partial = (float*)(c.ws + net->off_bnws);
                        fused_stat_rows = rows_needed;
                    }
                }
            } else {
                p.out = c.a(u.out);
                p.scale = c.bnc(u, 0); p.shift = c.bnc(u, 1);
                p.residual = u.res >= 0 ? c.a(u.res) : nullptr;
                p.relu = u.relu;
                // evaluation: this layer and the next one - its only reader, a plain conv + BN + ReLU - as ONE launch when both are
                // strip-kernel layers (smp's last decoder block at full resolution): the tensor between them is never written
                ensure_graph_maps(net);
                const int vi = unit_index + 1;
                if (vi < (int)net->units.size() && net->sole_consumer[u.out] == vi && u.res < 0 && u.bias_idx < 0 && u.bn_idx >= 0) {
                    const Unit& v = net->units[vi];
                    if (v.kind == U_CONV && v.src0 == u.out && v.src1 < 0 && !v.up0 && v.res < 0 && v.bn_idx >= 0 && v.gn_idx < 0 && v.bias_idx < 0 &&
                        !v.colr && !v.cg && !v.g2) {
                        ConvParams q = conv_params(c, v);
                        q.out = c.a(v.out); q.scale = c.bnc(v, 0); q.shift = c.bnc(v, 1); q.relu = v.relu;
                        ConvParams p1 = p;
                        p1.out = nullptr;
                        p1.out_f32 = 0; q.out_f32 = 0;
                        if (conv_pair_ok(dt, p1, q)) {
                            if ((rc = launch_conv_pair(dt, p1, q, c.s))) return rc;
                            prof_add_flops(conv_flops(c, v));
                            skip_units = 1;
                            break;
                        }
                    }
                }
            }
            if ((rc = launch_conv_igemm(dt, p, c.s))) return rc;
            break;
        }
        case U_ADD: {
            ProfScope prof(PK_POOL_MISC, 0, 3.0 * n * u.hout * u.wout * u.cout * net->esz, c.s);
            const int64_t rows = (int64_t)n * u.hout * u.wout;
            if ((rc = vs_channel_slice(dt, c.a(u.src0), u.cout, 0, c.a(u.out), u.cout, 0, u.cout, rows, 0, stream))) return rc;
            if ((rc = vs_channel_slice(dt, c.a(u.src1), u.cout, 0, c.a(u.out), u.cout, 0, u.cout, rows, 1, stream))) return rc;
            continue;
        }
        case U_CONVT: {   // 3x3 convolution onto 4 * cout channels, then the pixel shuffle (+ bias; eval: + folded BN + ReLU)
            ConvParams p = conv_params(c, u);
            p.out = c.ws + net->off_ct;
            {
                ProfScope prof(PK_CONV_FWD, conv_flops(c, u) * 4, 0, c.s);
                if ((rc = launch_conv_igemm(dt, p, c.s))) return rc;
            }
            ProfScope prof(PK_POOL_MISC, 0, 2.0 * n * u.hout * u.wout * u.cout * net->esz, c.s);
            if (training) {
                if ((rc = vs_depth_to_space2(dt, p.out, c.z(u.out), n, u.hin, u.win, u.cout, c.P(u.bias_idx), nullptr, nullptr, 0, stream))) return rc;
            } else {
                if ((rc = vs_depth_to_space2(dt, p.out, c.a(u.out), n, u.hin, u.win, u.cout, c.P(u.bias_idx), c.bnc(u, 0), c.bnc(u, 1), u.relu, stream))) return rc;
            }
            break;
        }
        case U_HEAD: {
            ConvParams p = conv_params(c, u);
            p.out = logits; p.shift = c.P(u.bias_idx); p.out_f32 = 3;  // fp32, NCHW
            ProfScope prof(PK_HEAD, conv_flops(c, u), 0, c.s);
            if (net->head_up > 1) {   // SegmentationHead(upsampling=4): the convolution at 1/4 resolution, then nn.UpsamplingBilinear2d
                p.out = c.ws + net->off_lsmall;
                if ((rc = launch_conv_igemm(dt, p, c.s))) return rc;
                if ((rc = vs_bilinear_up_planes((const float*)p.out, logits, n * net->classes, u.hout, u.wout, net->head_up, stream))) return rc;
                continue;
            }
            if (scatter) {
                p.scatter = scatter;
                if (conv_head_scatter_ok(dt, p)) {
                    p.out = nullptr;
                    if ((rc = launch_conv_igemm(dt, p, c.s))) return rc;
                    head_scattered = true;
                    continue;
                }
                p.scatter = nullptr;
            }
            if ((rc = launch_conv_igemm(dt, p, c.s))) return rc;
            continue;
        }
        }
        if (training) {  // batch statistics + normalise (+ residual) (+ ReLU)
            if (u.out >= 0 && net->nl_act[u.out]) continue;   // normalise-on-load: no sweep, no activation tensor
            if (fused_bins) {
                int64_t stat_rows = 0;
                if (net->stats_hook) {     // SyncBatchNorm: the integer sums of every rank, added in place (exact: the same bits everywhere)
                    const int hrc = net->stats_hook(net->stats_user, c.ws + u.off_bins, (int64_t)stat_bins_rows(u.cout) * 2 * u.cout, 0, (void*)c.s);
                    VS_REQUIRE(hrc == 0, "unet_forward: the cross-rank statistics hook failed (%d)", hrc);
                    stat_rows = c.rows(u) * net->stats_world;
                }
                ProfScope prof(PK_BN_APPLY, 0, act_bytes(c, u, u.res >= 0 ? 3 : 2), c.s);
                if ((rc = launch_bn_apply_from_bins(dt, c.z(u.out), (const unsigned long long*)(c.ws + u.off_bins), stat_bins_rows(u.cout), 1e-5f, 0.1f,
                                                    c.bnc(u, 2), c.bnc(u, 3), rm, rv, c.P(u.bn_idx), c.P(u.bn_idx + 1),
                                                    u.res >= 0 ? c.a(u.res) : nullptr, u.relu, c.a(u.out), c.rows(u), u.cout, c.s, stat_rows))) return rc;
                continue;
            }
            VS_REQUIRE(!net->stats_hook, "unet_forward: cross-rank BatchNorm statistics need the statistics bins for every unit (bf16, options stats_bins / fuse_stats)");
            if (fused_stat_rows && fused_stat_rows <= vs_option("bn_inline_rows")) {   // few partial rows: one launch does both
                ProfScope prof(PK_BN_APPLY, 0, act_bytes(c, u, u.res >= 0 ? 3 : 2), c.s);
                if ((rc = launch_bn_apply_from_partials(dt, c.z(u.out), (const float*)(c.ws + net->off_bnws), fused_stat_rows, 1e-5f, 0.1f,
                                                        c.bnc(u, 2), c.bnc(u, 3), rm, rv, c.P(u.bn_idx), c.P(u.bn_idx + 1),
                                                        u.res >= 0 ? c.a(u.res) : nullptr, u.relu, c.a(u.out), c.rows(u), u.cout, c.s))) return rc;
                continue;
            }
            if (fused_stat_rows) {
                ProfScope prof(PK_BN_STATS, 0, 0, c.s);
                if ((rc = launch_bn_finalize_partials((const float*)(c.ws + net->off_bnws), fused_stat_rows, u.cout, c.rows(u),
                                                      1e-5f, 0.1f, c.bnc(u, 2), c.bnc(u, 3), rm, rv, c.s))) return rc;
            } else {
                ProfScope prof(PK_BN_STATS, 0, act_bytes(c, u, 1), c.s);
                if ((rc = vs_bn_stats(dt, c.z(u.out), c.rows(u), u.cout, 1e-5f, 0.1f, c.bnc(u, 2), c.bnc(u, 3), rm, rv,
                                      (float*)(c.ws + net->off_bnws), net->bnws_bytes, stream))) return rc;
            }
            ProfScope prof(PK_BN_APPLY, 0, act_bytes(c, u, u.res >= 0 ? 3 : 2), c.s);
            if ((rc = vs_bn_apply(dt, c.z(u.out), c.bnc(u, 2), c.bnc(u, 3), c.P(u.bn_idx), c.P(u.bn_idx + 1),
                                  u.res >= 0 ? c.a(u.res) : nullptr, u.relu, c.a(u.out), c.rows(u), u.cout, stream))) return rc;
        }
    }
    return (scatter && !head_scattered) ? 1 : VS_OK;   // 1: the caller still has to turn the logits into volume entries
}

// ---- backward --------------------------------------------------------------------------------------
// role: which stream's share of the work a call enqueues.  ROLE_BOTH is the normal two-stream form (fork / join events
// inside).  ROLE_MAIN / ROLE_SIDE enqueue only the caller's-stream kernels / only the weight-gradient + optimiser kernels,
// both on `stream`: a recorded step keeps each share as its own LINEAR hipGraph (the runtime replays a linear graph of
// kernels as one batch of queue packets - 0.13 ms of host time per step - but walks a graph with parallel branches node by
// node, slower than launching call by call).  The split roles use no events: the caller orders the two shares itself, range
// by range (hipEventRecordWithFlags(hipEventRecordExternal), which would let ONE pair of graphs carry the fork events as
// external event nodes, returns hipErrorInvalidValue under capture on ROCm 7.2).
enum { ROLE_BOTH = 0, ROLE_MAIN = 1, ROLE_SIDE = 2 };
static int unet_backward_range(vs_unet_t* net, const float* params, const float* x, const float* dlogits, int n,
                               int need_encoder_wgrad, float* grads, void* workspace, void* stream, int unit_lo, int unit_hi,
                               const vs_adamw_args* opt, int role = ROLE_BOTH);

extern "C" int vs_unet_backward(vs_unet_t* net, const float* params, const float* x, const float* dlogits, int n,
                                int need_encoder_wgrad, float* grads, void* workspace, void* stream) {
    VS_REQUIRE(net, "unet_backward: null pointer");
    return unet_backward_range(net, params, x, dlogits, n, need_encoder_wgrad, grads, workspace, stream, 0, (int)net->units.size(), nullptr);
}

// Backward of the units [unit_lo, unit_hi) only (processed from unit_hi-1 down to unit_lo).  Calling it for consecutive
// ranges from the top of the network down is identical to one vs_unet_backward call; when it returns, every gradient of
// the range is ordered on the caller's stream (the side stream is joined), so a data-parallel caller can all-reduce that
// slice of the flat gradient buffer while the next range runs.
extern "C" int vs_unet_backward_range(vs_unet_t* net, const float* params, const float* x, const float* dlogits, int n,
                                      int need_encoder_wgrad, float* grads, void* workspace, void* stream, int unit_lo,
                                      int unit_hi) {
    VS_REQUIRE(net && unit_lo >= 0 && unit_lo < unit_hi && unit_hi <= (int)net->units.size(), "unet_backward_range: bad unit range");
    return unet_backward_range(net, params, x, dlogits, n, need_encoder_wgrad, grads, workspace, stream, unit_lo, unit_hi, nullptr);
}

// first tensor-table index (parameter-buffer element offset) owned by a unit: lets a caller map unit ranges to slices
extern "C" int64_t vs_unet_unit_param_offset(const vs_unet_t* net, int unit) {
    if (!net || unit < 0 || unit > (int)net->units.size()) return -1;
    if (unit == (int)net->units.size()) return net->layout.n_params;
    for (int u = unit; u < (int)net->units.size(); ++u)
        if (net->units[u].w_idx >= 0) return net->layout.tensors[net->units[u].w_idx].offset;
    return net->layout.n_params;
}

// AdamW over the parameter slices of the units [lo, hi) (everything, minus the frozen encoder convolutions when those are
// not trained) and the derived weight copies of their convolutions for the NEXT forward (the other weight set), on `s`.
static int update_units(const Ctx& c, int lo, int hi, bool need_encoder_wgrad, const float* grads, const vs_adamw_args& opt,
                        hipStream_t s) {
    vs_unet* net = c.net;
    const int dt = net->dtype;
    const int other = net->wset ^ 1;
    AdamwRanges r{};
    long w_off[64], wc_off[64], wt_off[64];
    int cout[64], taps[64], cin[64], cpad[64], cgs[64], upd[64], nl = 0;
    int rc;
    // Per range of units: ONE launch steps the small tensors (BatchNorm affine parameters, biases, the stem, depthwise / attention
    // tensors: ranges of the flat buffers), ONE launch steps the convolution weights INSIDE the derivation of their copies for the next
    // forward (round 4: the AdamW launch over the group's convolution slices and the copy launch were two; 0.025 ms per step)
    auto flush = [&]() -> int {
        int rc2;
        if (r.n && (rc2 = launch_adamw_ranges(opt, grads, r, s))) return rc2;
        if (nl && (rc2 = launch_adamw_prepare_all(dt, opt, grads, c.ws, nl, w_off, wc_off, wt_off, cout, taps, cin, cpad, cgs, upd, s))) return rc2;
        r.n = 0; nl = 0;
        return VS_OK;
    };
    auto push = [&](int idx) {
        const TensorInfo& t = c.t(idx);
        int64_t len = 1;
        for (int d = 0; d < t.ndim; ++d) len *= t.shape[d];
        if (r.n > 0 && r.off[r.n - 1] + r.len[r.n - 1] == t.offset) { r.len[r.n - 1] += len; return; }
        r.off[r.n] = t.offset; r.len[r.n] = len; ++r.n;
    };
    for (int k = lo; k < hi; ++k) {
        const Unit& v = net->units[k];
        if (v.w_idx < 0 && v.bn_idx < 0) continue;
        if (r.n > 150 || nl == 64) { if ((rc = flush())) return rc; }
        if (v.kind == U_SE) { push(v.w_idx); push(v.w_idx + 1); push(v.w_idx + 2); push(v.w_idx + 3); continue; }
        if (v.kind == U_FPA) { for (int t : v.tens) push(t); continue; }
        const bool conv_w = v.kind == U_CONV || v.kind == U_HEAD;          // its weight is updated by the fused copy launch below
        if (v.w_idx >= 0 && !conv_w && !(v.frozen_candidate && !need_encoder_wgrad)) push(v.w_idx);
        const bool aux_on = !(v.aux_frozen && !need_encoder_wgrad);      // (ResNeSt: BatchNorms / biases named conv2.bn0, conv2.fc1, conv1.1 ..)
        if (v.bn_idx >= 0 && aux_on) { push(v.bn_idx); push(v.bn_idx + 1); }
        if (v.gn_idx >= 0) { push(v.gn_idx); push(v.gn_idx + 1); }
        if (v.bias_idx >= 0 && aux_on) push(v.bias_idx);
        if (v.kind == U_CONV || v.kind == U_HEAD) {
            w_off[nl] = c.t(v.w_idx).offset;
            wc_off[nl] = (dt != VS_F32 || v.cg || v.g2) ? (long)Ctx::wc_off(v, other) : -1;
            wt_off[nl] = (long)Ctx::wt_off(v, other);
            cout[nl] = v.cout; taps[nl] = v.colr ? 1 : v.k * v.k; cin[nl] = v.colr ? 9 * v.cin0 : v.cin0 + v.cin1; cpad[nl] = v.kind == U_HEAD ? 16 : v.cout;
            cgs[nl] = v.g2 ? 255 : v.cg;
            upd[nl] = !(v.frozen_candidate && !need_encoder_wgrad);
            ++nl;
        }
    }
    if ((rc = flush())) return rc;
    for (int k = lo; k < hi; ++k) {   // transposed convolutions: their own expansion kernel, after every AdamW launch of the range
        const Unit& v = net->units[k];
        if (v.kind != U_CONVT) continue;
        if ((rc = launch_convt_weight_prepare(dt, opt.params + c.t(v.w_idx).offset, c.ws + Ctx::wc_off(v, other), c.ws + Ctx::wt_off(v, other),
                                              v.cin0, v.cout, s))) return rc;
    }
    return VS_OK;
}

static int unet_backward_range(vs_unet_t* net, const float* params, const float* x, const float* dlogits, int n,
                               int need_encoder_wgrad, float* grads, void* workspace, void* stream, int unit_lo, int unit_hi,
                               const vs_adamw_args* opt, int role) {
    VS_REQUIRE(net && params && x && dlogits && grads && workspace, "unet_backward: null pointer");
    VS_NO_F16(net->dtype, "unet_backward");
    const bool do_main = role != ROLE_SIDE, do_side = role != ROLE_MAIN;
    VS_REQUIRE(n == net->last_n, "unet_backward: batch %d does not match the last training forward (%d)", n, net->last_n);
    Ctx c{net, (char*)workspace, params, nullptr, (hipStream_t)stream, n};
    const int dt = net->dtype;
    int rc;
    if (do_main && unit_hi == (int)net->units.size()) net->written.assign(net->acts.size(), 0);  // a new backward pass starts at the top
    ensure_graph_maps(net);
    if (do_main && unit_hi == (int)net->units.size()) net->bwd_stat_rows.assign(net->acts.size(), 0);
    VS_REQUIRE(net->written.size() == net->acts.size(), "unet_backward_range: ranges must start at the last unit");
    std::vector<char>& written = net->written;
    float* wgws = (float*)(c.ws + net->off_wgws);
    const int n_side = role == ROLE_BOTH ? vs_option("side_stream") : 0;  // 0 = everything in order on the given stream
    const bool use_side = n_side > 0;
    hipStream_t ws_stream = c.s;  // stream of the current unit's weight-gradient work
    bool side_used[vs_unet::kSide] = {false, false};   // side streams this call forked onto (only those are joined: under
                                                       // stream capture a stream that never joined the capture must not be waited on)
    if (use_side) {
        // (a lowest-priority side stream was measured: 4.843 vs 4.853 ms per step - no preference of the dispatcher to speak of)
        if ((rc = acquire_side_streams(net, c.s))) return rc;
        for (int i = 0; i < vs_unet::kSide; ++i)
            if (!net->join_event[i]) VS_CHECK_HIP(hipEventCreateWithFlags(&net->join_event[i], hipEventDisableTiming));
        while (net->fork_events.size() < net->units.size()) {
            hipEvent_t e;
            VS_CHECK_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            net->fork_events.push_back(e);
        }
    }
    auto fork_mark = [&](int ui) -> int {  // everything enqueued on the caller's stream so far happens-before the side work
        if (!use_side) return VS_OK;
        VS_CHECK_HIP(hipEventRecord(net->fork_events[ui], c.s));
        return VS_OK;
    };
    auto fork_wait = [&](int ui) -> int {
        if (!use_side) return VS_OK;
        const int k = (n_side >= 2 && !opt) ? (ui & 1) : 0;   // the fused optimiser step relies on ONE side stream's order
        ws_stream = net->side[k];
        side_used[k] = true;
        wgws = (float*)(c.ws + net->off_wgws + (size_t)k * net->wgws_bytes);
        VS_CHECK_HIP(hipStreamWaitEvent(ws_stream, net->fork_events[ui], 0));
        return VS_OK;
    };
    // Fused optimiser step: the network is cut into groups (decoder + head, layer4, layer3, layer2, stem + layer1); when the
    // first unit of a group has queued its weight gradient, every gradient of the group is complete in side-stream order
    // (its BN / bias gradients were produced on the caller's stream before that unit's fork event), so ONE AdamW launch over
    // the group's parameter slice and ONE launch deriving its weight copies for the next forward follow on the side stream.
    auto group_update = [&](int ui) -> int {
        if (net->group_first.empty()) {   // unit index -> does an optimiser group start here
            static const char* const kCuts[] = {"decoder.", "encoder.layer4.0.", "encoder.layer3.0.", "encoder.layer2.0.", "encoder.layer1.0."};
            const int ncuts = 5;                                  // (the stem is a group of its own: layer1's update then runs under the max-pool /
            net->group_first.assign(net->units.size(), 0);        // stem BatchNorm backward instead of behind the stem's gradient - 0.03 ms per step)
            net->group_first[0] = 1;
            std::string prev;
            for (size_t k = 0; k < net->units.size(); ++k) {
                if (net->units[k].w_idx < 0) continue;
                const std::string& nm = net->layout.tensors[net->units[k].w_idx].name;
                for (int ci = 0; ci < ncuts; ++ci)
                    if (nm.rfind(kCuts[ci], 0) == 0 && prev.rfind(kCuts[ci], 0) != 0) net->group_first[k] = 1;
                prev = nm;
            }
        }
        if (!net->group_first[ui]) return VS_OK;
        int hi = ui + 1;   // the group is [ui, hi): up to the next group's first unit (or the end of the network)
        while (hi < (int)net->units.size() && !net->group_first[hi]) ++hi;
        return update_units(c, ui, hi, need_encoder_wgrad != 0, grads, *opt, ws_stream);
    };
    // Weight-gradient work of one unit, queued on the side stream (after a fork event that covers its dz).
    bool head_planes = false;                 // the head's data gradient reads dlogits' planes (head_dgrad_planes_kernel)
    const float* head_dl = dlogits;
    const bool head_planes_on = vs_option("conv_direct") != 0;    // (the strip-kernel family's switch)
    struct SideItem { int ui; const void* dzp; int dz_c; };
    std::vector<SideItem> pending;
    auto side_wgrad = [&](const SideItem& it) -> int {
        const int ui = it.ui;
        const Unit& u = net->units[ui];
        const void* dzp = it.dzp;
        const int dz_c = it.dz_c;
        int rc;
        const bool want_w = !(u.frozen_candidate && !need_encoder_wgrad);
        prof_set_tag(ui);
        if (u.kind == U_STEM) {
            ProfScope prof(PK_STEM, want_w ? 2.0 * n * u.hout * u.wout * 64 * 49 : 0, 0, ws_stream);
            if (want_w) {
                if ((rc = vs_stem_wgrad(dt, x, dzp, grads + c.t(u.w_idx).offset, wgws, net->wgws_bytes, n, net->h, net->w, (void*)ws_stream))) return rc;
            } else {
                VS_CHECK_HIP(hipMemsetAsync(grads + c.t(u.w_idx).offset, 0, 64 * 49 * sizeof(float), ws_stream));
            }
            return opt ? group_update(ui) : VS_OK;
        }
        if (u.kind == U_DWCONV2) {
            ProfScope prof(PK_CONV_WGRAD, want_w ? 2.0 * n * u.hout * u.wout * u.cout * u.k * u.k : 0, 0, ws_stream);
            if (want_w) {
                if ((rc = vs_dwconv2d_wgrad(dt, u.bcast ? (const void*)x : (const void*)c.a(u.src0), dzp, grads + c.t(u.w_idx).offset, n, u.hin, u.win, u.cout, u.k,
                                            u.stride, u.pad, u.dil, u.hout, u.wout, u.bcast, wgws, net->wgws_bytes, (void*)ws_stream))) return rc;
            } else {
                VS_CHECK_HIP(hipMemsetAsync(grads + c.t(u.w_idx).offset, 0, (size_t)u.cout * u.k * u.k * sizeof(float), ws_stream));
            }
            return opt ? group_update(ui) : VS_OK;
        }
        if (u.kind == U_DWCONV) {
            ProfScope prof(PK_CONV_WGRAD, 2.0 * n * u.hout * u.wout * u.cout * 9, 0, ws_stream);
            if ((rc = vs_dwconv3x3_wgrad(dt, c.a(u.src0), dzp, grads + c.t(u.w_idx).offset, n, u.hin, u.win, u.cout, u.dil, wgws, net->wgws_bytes,
                                         (void*)ws_stream))) return rc;
            return opt ? group_update(ui) : VS_OK;
        }
        if (want_w) {
            ProfScope prof(PK_CONV_WGRAD, conv_flops(c, u), 0, ws_stream);
            WgradParams p = wgrad_params(c, u);
            p.dy = dzp; p.Cout = dz_c;
            p.partials = wgws; p.partial_bytes = net->wgws_bytes;
            if (u.kind == U_CONVT) {   // dense gradient of the 3x3 form, then its 16 real taps into torch's [in][out][4][4]
                p.Hout = u.hin; p.Wout = u.win;
                const size_t k = ((const char*)wgws - (c.ws + net->off_wgws)) / net->wgws_bytes;
                p.dw = (float*)(c.ws + net->off_ctdw + k * net->ctdw_bytes);
                if ((rc = launch_conv_wgrad(dt, p, ws_stream))) return rc;
                if ((rc = launch_convt_wgrad_gather(p.dw, grads + c.t(u.w_idx).offset, u.cin0, u.cout, ws_stream))) return rc;
            } else if (u.g2) {        // dense gradient, then every output channel's own group half into the [cout][taps][cin / 2] tensor
                const size_t k = ((const char*)wgws - (c.ws + net->off_wgws)) / net->wgws_bytes;
                p.dw = (float*)(c.ws + net->off_ctdw + k * net->ctdw_bytes);
                if ((rc = launch_conv_wgrad(dt, p, ws_stream))) return rc;
                if ((rc = launch_two_group_wgrad_extract(p.dw, grads + c.t(u.w_idx).offset, u.cout, u.k * u.k, u.cin0, ws_stream))) return rc;
            } else if (u.kind == U_HEAD) {
                p.cout_live = net->classes;
                if (head_planes) {    // the bias gradient, and - unless the weight-gradient kernel reads the planes itself - its 16-channel operand: off the caller's stream
                    p.dy_planes = net->classes;
                    const bool direct = conv_wgrad_takes_planes(dt, p) && n <= 256;   // (the bias-only sweep's partial buffer: 4 segments x 256 images)
                    if (direct) p.dy = head_dl; else p.dy_planes = 0;
                    if ((rc = launch_dlogits_to_nhwc16(dt, head_dl, direct ? nullptr : c.ws + net->off_dyh, n, net->classes, (int64_t)u.hout * u.wout,
                                                       grads + c.t(u.bias_idx).offset, (float*)(c.ws + net->off_headpart), ws_stream))) return rc;
                }
                if (conv_wgrad_honours_cout_live(dt, p)) {   // the row-streaming kernel leaves [classes][9][16] slabs: summed straight into the gradient
                    p.dw = grads + c.t(u.w_idx).offset;
                    if ((rc = launch_conv_wgrad(dt, p, ws_stream))) return rc;
                } else {
                    p.cout_live = 0;
                    p.dw = (float*)(c.ws + net->off_headdw);
                    if ((rc = launch_conv_wgrad(dt, p, ws_stream))) return rc;
                    VS_CHECK_HIP(hipMemcpyAsync(grads + c.t(u.w_idx).offset, p.dw, (size_t)net->classes * u.k * u.k * u.cin0 * sizeof(float),
                                                hipMemcpyDeviceToDevice, ws_stream));
                }
            } else {
                // (the split-K slabs are summed right behind the kernel, out of the ONE slab buffer every layer reuses: it stays in the
                // L2s / Infinity Cache.  Round 4 measured the alternatives: slabs kept per layer and summed once per optimiser group
                // (+0.02 ms per step: 0.7 GB of slabs then go to HBM and back) or inside the optimiser launch (+0.33 ms))
                p.dw = grads + c.t(u.w_idx).offset;
                if ((rc = launch_conv_wgrad(dt, p, ws_stream))) return rc;
            }
        } else {
            VS_CHECK_HIP(hipMemsetAsync(grads + c.t(u.w_idx).offset, 0,
                                        (size_t)u.cout * u.k * u.k * (u.cg ? u.cg : (u.cin0 + u.cin1) / (u.g2 ? 2 : 1)) * sizeof(float), ws_stream));
        }
        return opt ? group_update(ui) : VS_OK;
    };
    const int fork_every = std::max(1, vs_option("fork_every"));
    const bool stem_on_caller = use_side && opt && role == ROLE_BOTH;
    for (int ui = unit_hi - 1; ui >= unit_lo; --ui) {
        const Unit& u = net->units[ui];
        prof_set_tag(ui);
        if (u.kind == U_CONCAT) {   // gradient of the concatenation: each slice is added onto its member's gradient
            if (!do_main) continue;
            VS_REQUIRE(written[u.out], "backward: concat gradient missing");
            ProfScope prof(PK_POOL_MISC, 0, 3.0 * n * u.hout * u.wout * u.cout * net->esz, c.s);
            int off = 0;
            for (int m : u.members) {
                const int mc = net->acts[m].c;
                if ((rc = vs_channel_slice(dt, c.da(u.out), u.cout, off, c.da(m), mc, 0, mc, (int64_t)n * u.hout * u.wout, written[m], stream))) return rc;
                written[m] = 1;
                off += mc;
            }
            continue;
        }
        if (u.kind == U_ADD) {      // both addends receive the sum's gradient
            if (!do_main) continue;
            VS_REQUIRE(written[u.out], "backward: gradient of a sum missing");
            ProfScope prof(PK_POOL_MISC, 0, 4.0 * n * u.hout * u.wout * u.cout * net->esz, c.s);
            for (int m : {u.src0, u.src1}) {
                if ((rc = vs_channel_slice(dt, c.da(u.out), u.cout, 0, c.da(m), u.cout, 0, u.cout, (int64_t)n * u.hout * u.wout, written[m], stream))) return rc;
                written[m] = 1;
            }
            continue;
        }
        if (u.kind == U_DROPADD) {  // the skip takes the gradient as is, the block's branch the surviving samples' (scaled by 1 / keep)
            if (!do_main) continue;
            VS_REQUIRE(written[u.out] && !written[u.src0], "backward: gradients of an MBConv skip out of order");
            ProfScope prof(PK_POOL_MISC, 0, 4.0 * n * u.hout * u.wout * u.cout * net->esz, c.s);
            if ((rc = vs_channel_slice(dt, c.da(u.out), u.cout, 0, c.da(u.src1), u.cout, 0, u.cout, (int64_t)n * u.hout * u.wout, written[u.src1], stream))) return rc;
            written[u.src1] = 1;
            if ((rc = vs_sample_scale_add(dt, c.da(u.out), u.drop_p > 0.f ? (const float*)(c.ws + u.off_gn) : nullptr, nullptr, c.da(u.src0), n,
                                          (int64_t)u.hout * u.wout * u.cout, stream))) return rc;
            written[u.src0] = 1;
            continue;
        }
        if (u.kind == U_BN) {       // dx, dgamma, dbeta; the activation's derivative is recomputed from the pre-norm tensor
            if (!do_main) continue;
            VS_REQUIRE(written[u.out] && !written[u.src0], "backward: gradients of a BatchNorm unit out of order");
            ProfScope prof(PK_BN_BWD, 0, 5.0 * n * u.hout * u.wout * u.cout * net->esz, c.s);
            if ((rc = vs_bn2_bwd(dt, c.da(u.out), c.a(u.src0), c.bnc(u, 2), c.bnc(u, 3), c.P(u.bn_idx), c.P(u.bn_idx + 1), u.relu, c.da(u.src0),
                                 grads + c.t(u.bn_idx).offset, grads + c.t(u.bn_idx + 1).offset, (int64_t)n * u.hout * u.wout, u.cout,
                                 (float*)(c.ws + net->off_bnws), net->bnws_bytes, stream))) return rc;
            written[u.src0] = 1;
            continue;
        }
        if (u.kind == U_SIGMOID) {
            if (!do_main) continue;
            VS_REQUIRE(written[u.out] && !written[u.src0], "backward: sigmoid gradients out of order");
            if ((rc = vs_sigmoid_bwd(dt, c.da(u.out), c.a(u.out), c.da(u.src0), (int64_t)n * u.hout * u.wout * u.cout, stream))) return rc;
            written[u.src0] = 1;
            continue;
        }
        if (u.kind == U_FPA) {      // everything on the caller's stream (tiny): combination, pooled branch, pyramid, 7x7 convolution, max-pool
            if (!do_main) continue;
            VS_REQUIRE(written[u.out] && !written[u.src1] && !written[u.res], "backward: FPA gradients out of order");
            ProfScope prof(PK_POOL_MISC, 0, 4.0 * n * u.hin * u.win * u.cin0 * net->esz, c.s);
            float* arena = (float*)(c.ws + u.off_fpa_arena);
            float* plane = (float*)(c.ws + u.off_fpa_plane);
            float* dplane = plane + (size_t)n * u.hin * u.win;
            if ((rc = vs_fpa_combine_bwd(dt, c.da(u.out), plane, c.a(u.src1), c.da(u.src1), dplane, n, (int64_t)u.hout * u.wout, u.cout, stream))) return rc;
            if ((rc = vs_spatial_sum(dt, c.da(u.out), c.da(u.res), n, (int64_t)u.hout * u.wout, u.cout, 1.f, stream))) return rc;
            written[u.src1] = 1; written[u.res] = 1;
            float* pp[36]; float* gp[24];
            for (int i = 0; i < 6; ++i) {
                for (int j = 0; j < 4; ++j) { pp[6 * i + j] = const_cast<float*>(c.P(u.tens[4 * i + j])); gp[4 * i + j] = grads + c.t(u.tens[4 * i + j]).offset; }
                pp[6 * i + 4] = nullptr; pp[6 * i + 5] = nullptr;       // running statistics are not touched by the backward pass
            }
            if ((rc = vs_fpa_pyramid_bwd(arena, dplane, pp, gp, n, u.hin, u.win, stream))) return rc;
            // down1's convolution: its weight / bias gradients overwrite slots 0 / 1 (the pyramid kernel leaves them alone)
            char* dpool = c.ws + u.off_fpa_pool + (size_t)n * (u.hin / 2) * (u.win / 2) * u.cin0 * net->esz;
            if ((rc = vs_conv_to_plane_bwd(dt, c.ws + u.off_fpa_pool, c.P(u.tens[0]), arena + vs_fpa_dz1_offset(n, u.hin, u.win), dpool,
                                           grads + c.t(u.tens[0]).offset, grads + c.t(u.tens[1]).offset, n, u.hin / 2, u.win / 2, u.cin0, 7, stream))) return rc;
            if ((rc = vs_maxpool2x2_bwd(dt, c.a(u.src0), dpool, c.da(u.src0), n, u.hin, u.win, u.cin0, written[u.src0] ? 1 : 0, stream))) return rc;
            written[u.src0] = 1;
            continue;
        }
        if (u.kind == U_PAB) {      // identity path + the attention term's gradients w.r.t. its three convolution outputs
            if (!do_main) continue;
            VS_REQUIRE(written[u.out] && !written[u.members[0]] && !written[u.members[1]] && !written[u.members[2]], "backward: PAB gradients out of order");
            ProfScope prof(PK_POOL_MISC, 0, 6.0 * n * u.hout * u.wout * u.cout * net->esz, c.s);
            if ((rc = vs_channel_slice(dt, c.da(u.out), u.cout, 0, c.da(u.src0), u.cout, 0, u.cout, (int64_t)n * u.hout * u.wout, written[u.src0], stream))) return rc;
            written[u.src0] = 1;
            if ((rc = vs_pab_attention_bwd(dt, c.da(u.out), c.a(u.members[0]), c.a(u.members[1]), c.a(u.members[2]), (const float*)(c.ws + u.off_gn),
                                           c.da(u.members[0]), c.da(u.members[1]), c.da(u.members[2]), (float*)(c.ws + net->off_pab), n,
                                           u.hout * u.wout, u.cin0, u.cout, stream))) return rc;
            for (int m : u.members) written[m] = 1;
            continue;
        }
        if (u.kind == U_SE) {       // the gate's parameter gradients are produced here, on the caller's stream (a few hundred values)
            if (!do_main) continue;
            VS_REQUIRE(written[u.out] && !written[u.src0], "backward: SE gate gradients out of order");
            ProfScope prof(PK_POOL_MISC, 12.0 * n * u.cout * u.cin1, 0, c.s);
            if ((rc = vs_se_gate_bwd(dt, c.da(u.out), c.a(u.out), c.a(u.src0), (const float*)(c.ws + u.off_gn), c.P(u.w_idx), c.P(u.w_idx + 2),
                                     c.da(u.src0), grads + c.t(u.w_idx).offset, grads + c.t(u.w_idx + 1).offset, grads + c.t(u.w_idx + 2).offset,
                                     grads + c.t(u.w_idx + 3).offset, (float*)(c.ws + net->off_sews), n, u.cout, u.cin1, u.relu == 2, stream))) return rc;
            written[u.src0] = 1;
            continue;
        }
        if (u.kind == U_CGATE) {    // d(input) = dy * gate, d(gate)[n][c] = sum over the map of dy * input
            if (!do_main) continue;
            VS_REQUIRE(written[u.out] && !written[u.src0] && !written[u.src1], "backward: channel gate gradients out of order");
            ProfScope prof(PK_POOL_MISC, 0, 4.0 * n * u.hout * u.wout * u.cout * net->esz, c.s);
            if ((rc = vs_sample_rowsum_ws(dt, c.a(u.src0), c.da(u.out), c.da(u.src1), n, (int64_t)u.hout * u.wout, u.cout, 1.f,
                                          (float*)(c.ws + net->off_gapws), net->gapws_bytes, stream))) return rc;
            if ((rc = vs_channel_gate(dt, c.da(u.out), c.a(u.src1), c.da(u.src0), n, (int64_t)u.hout * u.wout, u.cout, stream))) return rc;
            written[u.src0] = 1; written[u.src1] = 1;
            continue;
        }
        if (u.kind == U_GAP) {      // every position receives the pooled gradient / hw
            if (!do_main) continue;
            VS_REQUIRE(written[u.out], "backward: gradient of a pooled feature missing");
            ProfScope prof(PK_POOL_MISC, 0, 2.0 * n * u.hin * u.win * u.cout * net->esz, c.s);
            if ((rc = vs_broadcast_rows(dt, c.da(u.out), c.da(u.src0), n, (int64_t)u.hin * u.win, u.cout, 1.f / (float)(u.hin * u.win),
                                        written[u.src0] ? 1 : 0, stream))) return rc;
            written[u.src0] = 1;
            continue;
        }
        if (u.kind == U_BCAST) {    // the 1x1 map receives the sum over the positions it was copied to
            if (!do_main) continue;
            VS_REQUIRE(written[u.out] && !written[u.src0], "backward: broadcast gradient missing / its source written twice");
            ProfScope prof(PK_POOL_MISC, 0, (double)n * u.hout * u.wout * u.cout * net->esz, c.s);
            if ((rc = vs_spatial_sum(dt, c.da(u.out), c.da(u.src0), n, (int64_t)u.hout * u.wout, u.cout, 1.f, stream))) return rc;
            written[u.src0] = 1;
            continue;
        }
        if (u.kind == U_DROPOUT_E) {
            if (!do_main) continue;
            VS_REQUIRE(written[u.out] && !written[u.src0], "backward: dropout gradient missing / its input's gradient already written");
            ProfScope prof(PK_POOL_MISC, 0, 2.0 * n * u.hout * u.wout * u.cout * net->esz, c.s);
            if ((rc = vs_dropout(dt, c.da(u.out), c.da(u.src0), (int64_t)n * u.hout * u.wout * u.cout, 0.5f, net->rng_seed ^ 0x5bd1e995u,
                                 net->rng_counter, 0, stream))) return rc;
            written[u.src0] = 1;
            continue;
        }
        if (u.kind == U_FOLD2) {    // both splits receive the sum's gradient
            if (!do_main) continue;
            VS_REQUIRE(written[u.out], "backward: gradient of a radix sum missing");
            ProfScope prof(PK_POOL_MISC, 0, 3.0 * n * u.hout * u.wout * u.cout * net->esz, c.s);
            const int64_t rows = (int64_t)n * u.hout * u.wout;
            const int acc = written[u.src0] ? 1 : 0;
            if ((rc = vs_channel_slice(dt, c.da(u.out), u.cout, 0, c.da(u.src0), 2 * u.cout, 0, u.cout, rows, acc, stream))) return rc;
            if ((rc = vs_channel_slice(dt, c.da(u.out), u.cout, 0, c.da(u.src0), 2 * u.cout, u.cout, u.cout, rows, acc, stream))) return rc;
            written[u.src0] = 1;
            continue;
        }
        if (u.kind == U_RADIXSUM) { // d(splits) = dout * gate per split; d(gate)[n][r c + ch] = sum over the map of dout * split r
            if (!do_main) continue;
            VS_REQUIRE(written[u.out] && !written[u.src0] && !written[u.src1], "backward: gradients of a radix sum out of order");
            ProfScope prof(PK_POOL_MISC, 0, 6.0 * n * u.hout * u.wout * u.cout * net->esz, c.s);
            if ((rc = vs_sample_rowsum_b(dt, c.a(u.src0), c.da(u.out), u.cout, c.da(u.src1), n, (int64_t)u.hout * u.wout, 2 * u.cout,
                                         (float*)(c.ws + net->off_gapws), net->gapws_bytes, stream))) return rc;
            if ((rc = vs_radix2_gated_sum_bwd(dt, c.da(u.out), c.a(u.src1), c.da(u.src0), n, (int64_t)u.hout * u.wout, u.cout, stream))) return rc;
            written[u.src0] = 1; written[u.src1] = 1;
            continue;
        }
        if (u.kind == U_RSOFTMAX) {
            if (!do_main) continue;
            VS_REQUIRE(written[u.out] && !written[u.src0], "backward: radix softmax gradients out of order");
            if ((rc = vs_radix2_softmax_bwd(dt, c.da(u.out), c.a(u.out), c.da(u.src0), n, u.cout / 2, stream))) return rc;
            written[u.src0] = 1;
            continue;
        }
        if (u.kind == U_AVGPOOL) {  // the adjoint of the constant-tap depthwise convolution
            if (!do_main) continue;
            VS_REQUIRE(written[u.out], "backward: gradient of an average pool missing");
            ProfScope prof(PK_POOL_MISC, 0, (double)n * (u.hin * u.win + u.hout * u.wout) * u.cout * net->esz, c.s);
            const float* taps = (const float*)(c.ws + (u.k == 3 ? net->off_avgw9 : net->off_avgw4));
            if ((rc = vs_dwconv2d_bwd_data(dt, c.da(u.out), taps, c.da(u.src0), n, u.hin, u.win, u.cout, u.k, u.stride, u.pad, 1, u.hout, u.wout,
                                           written[u.src0] ? 1 : 0, stream))) return rc;
            written[u.src0] = 1;
            continue;
        }
        if (u.kind == U_UP2) {      // the 2x2 sums of the upsampled tensor's gradient
            if (!do_main) continue;
            VS_REQUIRE(written[u.out], "backward: gradient of an upsampled tensor missing");
            ProfScope prof(PK_POOL_MISC, 0, 1.25 * n * u.hout * u.wout * u.cout * net->esz, c.s);
            if ((rc = launch_upsample2x_bwd(dt, c.da(u.out), c.da(u.src0), n, u.hin, u.win, u.cout, written[u.src0] ? 1 : 0, c.s))) return rc;
            written[u.src0] = 1;
            continue;
        }
        if (u.kind == U_UPADD) {    // the lateral addend takes the gradient as is, the upsampled one its 2x2 sums
            if (!do_main) continue;
            VS_REQUIRE(written[u.out], "backward: gradient of a pyramid level missing");
            ProfScope prof(PK_POOL_MISC, 0, 3.5 * n * u.hout * u.wout * u.cout * net->esz, c.s);
            if ((rc = vs_channel_slice(dt, c.da(u.out), u.cout, 0, c.da(u.src1), u.cout, 0, u.cout, (int64_t)n * u.hout * u.wout, written[u.src1], stream))) return rc;
            written[u.src1] = 1;
            if ((rc = launch_upsample2x_bwd(dt, c.da(u.out), c.da(u.src0), n, u.hin, u.win, u.cout, written[u.src0] ? 1 : 0, c.s))) return rc;
            written[u.src0] = 1;
            continue;
        }
        if (u.kind == U_BILINEAR) {
            if (!do_main) continue;
            VS_REQUIRE(written[u.out], "backward: gradient of an upsampled tensor missing");
            ProfScope prof(PK_POOL_MISC, 0, 1.25 * n * u.hout * u.wout * u.cout * net->esz, c.s);
            if ((rc = vs_bilinear_up_bwd(dt, c.da(u.out), c.da(u.src0), n, u.hin, u.win, u.cout, u.factor, written[u.src0] ? 1 : 0, stream))) return rc;
            written[u.src0] = 1;
            continue;
        }
        if (u.kind == U_DROPOUT) {
            if (!do_main) continue;
            VS_REQUIRE(written[u.out] && !written[u.src0], "backward: dropout gradient missing / its input's gradient already written");
            ProfScope prof(PK_POOL_MISC, 0, 2.0 * n * u.hout * u.wout * u.cout * net->esz, c.s);
            if ((rc = vs_channel_scale(dt, c.da(u.out), (const float*)(c.ws + net->off_dropmask), c.da(u.src0), n, (int64_t)u.hout * u.wout, u.cout, stream))) return rc;
            written[u.src0] = 1;
            continue;
        }
        if (u.kind == U_POOL) {
            if (!do_main) continue;
            VS_REQUIRE(written[u.out], "backward: pool output gradient missing");
            ProfScope prof(PK_POOL_MISC, 0, (double)n * u.hin * u.win * 64 * net->esz * 1.5, c.s);
            if ((rc = vs_maxpool_bwd(dt, c.da(u.out), (const uint8_t*)(c.ws + net->off_idx), c.da(u.src0), written[u.src0], n,
                                     u.hin, u.win, u.cout, stream))) return rc;
            written[u.src0] = 1;
            continue;
        }
        const void* dzp;  // gradient w.r.t. the conv output of this unit
        int dz_c;
        if (u.kind == U_HEAD) {
            void* dyh = c.ws + net->off_dyh;
            // The data gradient reads dLoss / dlogits straight from its planes where head_dgrad_planes_kernel applies; the 16-channel NHWC
            // form (+ the bias gradient of the same sweep) is then only the weight gradient's operand and is made on ITS stream.
            head_planes = head_planes_on && u.k == 3 && u.stride == 1 && u.pad == 1 && u.dil <= 1 && !u.cg && u.cin1 == 0 && !u.up0 &&
                          net->first_consumer[u.src0] == ui && head_dgrad_planes_ok(dt, net->classes, u.hout, u.wout, u.cin0);
            head_dl = dlogits;
            if (do_main) {
            ProfScope prof(PK_HEAD, 0, (double)n * net->h * net->w * (net->classes * 8 + 16 * net->esz), c.s);
            if (net->head_up > 1) {   // back through the head's bilinear upsampling first
                float* ds = (float*)(c.ws + net->off_dlsmall);
                if ((rc = vs_bilinear_up_planes_bwd(dlogits, ds, n * net->classes, u.hout, u.wout, net->head_up, stream))) return rc;
            }
            }
            if (net->head_up > 1) head_dl = (const float*)(c.ws + net->off_dlsmall);
            if (do_main && !head_planes) {
            ProfScope prof(PK_HEAD, 0, (double)n * net->h * net->w * (net->classes * 8 + 16 * net->esz), c.s);
            if ((rc = launch_dlogits_to_nhwc16(dt, head_dl, dyh, n, net->classes, (int64_t)u.hout * u.wout, grads + c.t(u.bias_idx).offset,
                                               (float*)(c.ws + net->off_bnws), c.s))) return rc;   // + bias gradient, same sweep
            }
            dzp = dyh; dz_c = 16;
        } else if (!do_main) {
            dzp = c.dz(u.out); dz_c = u.cout;
            if (u.kind == U_CONVT) { dzp = c.da(u.out); dz_c = 4 * u.cout; }
            if (u.kind == U_CONV && u.bn_idx < 0 && u.gn_idx < 0) dzp = c.da(u.out);
            if (u.kind == U_DWCONV || u.kind == U_DWCONV2) dzp = c.da(u.out);
        } else if (u.kind == U_DWCONV || u.kind == U_DWCONV2) {      // no norm, no activation: dz IS the output's gradient
            VS_REQUIRE(written[u.out], "backward: gradient of unit %d output missing", ui);
            dzp = c.da(u.out); dz_c = u.cout;
        } else if (u.kind == U_CONV && u.bn_idx < 0 && u.gn_idx < 0) {   // plain biased convolution: dz IS the output's gradient
            VS_REQUIRE(written[u.out], "backward: gradient of unit %d output missing", ui);
            ProfScope prof(PK_POOL_MISC, 0, (double)n * u.hout * u.wout * u.cout * net->esz, c.s);
            if (u.bias_idx >= 0 &&
                (rc = vs_colsum(dt, c.da(u.out), c.rows(u), u.cout, grads + c.t(u.bias_idx).offset, (float*)(c.ws + net->off_bnws), net->bnws_bytes, stream))) return rc;
            dzp = c.da(u.out); dz_c = u.cout;
        } else if (u.kind == U_CONV && u.gn_idx >= 0) {                  // GroupNorm + ReLU backward
            VS_REQUIRE(written[u.out], "backward: gradient of unit %d output missing", ui);
            ProfScope prof(PK_BN_BWD, 0, act_bytes(c, u, 5), c.s);
            if ((rc = vs_gn_bwd(dt, c.da(u.out), c.z(u.out), (const float*)(c.ws + u.off_gn), c.P(u.gn_idx), c.P(u.gn_idx + 1), u.relu, c.dz(u.out),
                                grads + c.t(u.gn_idx).offset, grads + c.t(u.gn_idx + 1).offset, n, (int64_t)u.hout * u.wout, u.cout, u.gn_groups,
                                (float*)(c.ws + net->off_gnws), net->gnws_bytes, stream))) return rc;
            dzp = c.dz(u.out); dz_c = u.cout;
        } else {
            VS_REQUIRE(written[u.out], "backward: gradient of unit %d output missing", ui);
            void* dres = nullptr;
            if (u.res >= 0) {
                VS_REQUIRE(!written[u.res], "backward: residual gradient written twice (unit %d)", ui);
                dres = c.da(u.res);
                written[u.res] = 1;
            }
            BnSync sy{net->stats_hook, net->stats_user, net->stats_world, (float*)(c.ws + net->off_syncsc)};
            const BnSync* sync = net->stats_hook ? &sy : nullptr;
            if (net->bwd_stat_rows[u.out] > 0) {
                // the dgrad that completed da(u.out) left the masked gradient and the reduction's partial rows: one sweep
                ProfScope prof(PK_BN_BWD, 0, act_bytes(c, u, 3 + (dres ? 1 : 0)), c.s);
                if ((rc = launch_bn_bwd_from_partials(dt, c.da(u.out), c.z(u.out), c.bnc(u, 2), c.bnc(u, 3), c.P(u.bn_idx), c.dz(u.out), dres,
                                                      grads + c.t(u.bn_idx).offset, grads + c.t(u.bn_idx + 1).offset, c.rows(u), u.cout,
                                                      (const float*)(c.ws + net->off_bnws), net->bwd_stat_rows[u.out], c.s, sync))) return rc;
                net->bwd_stat_rows[u.out] = 0;
            } else {
            // two sweeps: (dy, y, x) reduce, then (dy, y, x) -> dx (+ dres)
            const bool recompute_mask = u.relu && u.res < 0;  // mask from x: two tensor reads fewer
            ProfScope prof(PK_BN_BWD, 0, act_bytes(c, u, ((u.relu && !recompute_mask) ? 6 : 4) + 1 + (dres ? 1 : 0)), c.s);
            if ((rc = bn_bwd_dispatch(dt, c.da(u.out), recompute_mask ? nullptr : c.a(u.out), c.z(u.out), c.bnc(u, 2),
                                      c.bnc(u, 3), c.P(u.bn_idx), c.P(u.bn_idx + 1), u.relu, c.dz(u.out), dres,
                                      grads + c.t(u.bn_idx).offset, grads + c.t(u.bn_idx + 1).offset, c.rows(u), u.cout,
                                      (float*)(c.ws + net->off_bnws), net->bnws_bytes, (unsigned*)(c.ws + net->off_bncnt), c.s, sync))) return rc;
            }
            dzp = c.dz(u.out); dz_c = u.cout;
            if (u.kind == U_CONV && u.bias_idx >= 0) {   // a biased convolution in front of BatchNorm (smp's ConvBnRelu): column sums of dz
                if ((rc = vs_colsum(dt, dzp, c.rows(u), u.cout, grads + c.t(u.bias_idx).offset, (float*)(c.ws + net->off_bnws), net->bnws_bytes, stream))) return rc;
            }
            if (u.kind == U_CONVT) {
                // the bias gradient (column sums of dz), then dz back through the pixel shuffle: the gradient of the 3x3 form's
                // output, kept in the (now dead) da buffer of this unit - the side stream's weight gradient reads it later
                ProfScope prof(PK_POOL_MISC, 0, 3.0 * n * u.hout * u.wout * u.cout * net->esz, c.s);
                if ((rc = vs_colsum(dt, dzp, c.rows(u), u.cout, grads + c.t(u.bias_idx).offset, (float*)(c.ws + net->off_bnws), net->bnws_bytes, stream))) return rc;
                if ((rc = vs_space_to_depth2(dt, dzp, c.da(u.out), n, u.hin, u.win, u.cout, stream))) return rc;
                dzp = c.da(u.out); dz_c = 4 * u.cout;
            }
        }
        // ---- fork policy: one event (a barrier packet on the caller's stream, ~7 us of command-processor time) covers the
        // weight-gradient work of up to `fork_every` consecutive units ----
        pending.push_back(SideItem{ui, dzp, dz_c});
        // (releasing the head's weight gradient at once - its operands exist before the backward pass starts - was measured: 4.512 vs 4.477 ms, not kept)
        const bool flush = (int)pending.size() >= fork_every || ui == unit_lo || u.kind == U_STEM;
        // (joining the side stream after every unit, or releasing a weight gradient only behind its unit's data gradient, were both
        // measured slower - 6.0 ms per step - and are not kept)
        if (flush && do_main && (rc = fork_mark(ui))) return rc;   // dz of every pending unit is complete at this point of the caller's stream
        if (u.colr && do_main) {   // data gradient of the 1x1 form = the input gradient's column form; its adjoint scatter onto the map
            ConvParams p{};
            p.src0 = dzp; p.C0 = u.cout; p.N = n; p.Hin = p.Hout = u.hin; p.Win = p.Wout = u.win; p.stride = 1; p.pad = 0; p.KH = p.KW = 1;
            p.w = c.ws + Ctx::wt_off(u, net->wset); p.Cout = 9 * u.cin0; p.out = c.ws + net->off_ys;
            {
                ProfScope prof(PK_CONV_DGRAD, conv_flops(c, u), 0, c.s);
                if ((rc = launch_conv_igemm(dt, p, c.s))) return rc;
            }
            if ((rc = vs_dilated_im2col(dt, p.out, c.da(u.src0), n, u.hin, u.win, u.cin0, u.colr, 1, written[u.src0] ? 1 : 0, stream))) return rc;
            written[u.src0] = 1;
        } else if (u.kind == U_DWCONV2 && do_main) {  // data gradient of a strided / padded depthwise convolution (none for the stem: the input)
            if (!u.bcast) {
                ProfScope prof(PK_CONV_DGRAD, 2.0 * n * u.hout * u.wout * u.cout * u.k * u.k, 0, c.s);
                if ((rc = vs_dwconv2d_bwd_data(dt, dzp, c.P(u.w_idx), c.da(u.src0), n, u.hin, u.win, u.cout, u.k, u.stride, u.pad, u.dil, u.hout, u.wout,
                                               written[u.src0] ? 1 : 0, stream))) return rc;
                written[u.src0] = 1;
            }
        } else if (u.kind == U_DWCONV && do_main) {   // data gradient of a depthwise convolution: the same sweep, taps reversed
            ProfScope prof(PK_CONV_DGRAD, 2.0 * n * u.hout * u.wout * u.cout * 9, 0, c.s);
            if ((rc = vs_dwconv3x3(dt, dzp, c.P(u.w_idx), c.da(u.src0), n, u.hin, u.win, u.cout, u.dil, 1 | (written[u.src0] ? 2 : 0), stream))) return rc;
            written[u.src0] = 1;
        } else if (u.kind == U_HEAD && head_planes && do_main) {
            ProfScope prof(PK_CONV_DGRAD, 2.0 * n * u.hout * u.wout * net->classes * u.k * u.k * u.cin0, 0, c.s);
            if ((rc = launch_head_dgrad_planes(dt, head_dl, c.ws + Ctx::wc_off(u, net->wset), c.da(u.src0), n, net->classes, u.hout, u.wout, u.cin0, c.s))) return rc;
            written[u.src0] = 1;
        } else if (u.kind != U_STEM && do_main) {
        // ---- data gradient (queued before the side-stream work so the caller's stream is fed first) ----
        ConvParams p{};
        const void* dsrc = dzp;
        const bool stuff_in_loader = u.stride == 2 && !(u.hin & 1) && !(u.win & 1);
        if (u.stride == 2 && !stuff_in_loader) {
            void* zs = c.ws + net->off_zs;
            ProfScope prof(PK_POOL_MISC, 0, (double)n * u.hin * u.win * dz_c * net->esz * 1.25, c.s);
            if ((rc = vs_zero_stuff2x(dt, dzp, zs, n, u.hout, u.wout, dz_c, stream))) return rc;
            dsrc = zs;
        }
        p.src0 = dsrc; p.C0 = dz_c; p.N = n; p.Hin = u.hin; p.Win = u.win; p.Hout = u.hin; p.Wout = u.win;
        if (stuff_in_loader) p.up0 = 2;   // the patch loader reads dz at the even positions and zeros elsewhere
        p.stride = 1; p.pad = u.pad; p.KH = p.KW = u.k;
        p.w = c.ws + Ctx::wt_off(u, net->wset); p.Cout = u.cin0 + u.cin1;
        p.gc = u.cg ? 32 : 0;
        p.dil = u.dil;
        if (u.up0) {
            // U-Net: every decoder input has this one consumer.  U-Net++: the upsampled input may already hold the contributions
            // of the nodes that read it as a dense skip - then the 2x2 sum goes through the separate, accumulating kernel.
            const bool acc0 = written[u.src0] != 0;
            VS_REQUIRE(u.src1 < 0 || !written[u.src1], "backward: decoder skip gradient written twice");
            if (u.src1 >= 0) { p.out1 = c.da(u.src1); p.split_c = u.cin0; written[u.src1] = 1; }
            if (!acc0 && conv_igemm_can_pool(p)) {  // 2x2 sum of the upsampled part inside the dgrad epilogue
                p.pool0 = 1;
                p.out = c.da(u.src0);
                prof_set_variant(conv_igemm_variant(dt, p));
                ProfScope prof(PK_CONV_DGRAD, conv_flops(c, u), 0, c.s);
                prof_set_variant(0);
                if ((rc = launch_conv_igemm(dt, p, c.s))) return rc;
            } else {
                p.out = c.ws + net->off_dup;
                {
                    ProfScope prof(PK_CONV_DGRAD, conv_flops(c, u), 0, c.s);
                    if ((rc = launch_conv_igemm(dt, p, c.s))) return rc;
                }
                ProfScope prof(PK_POOL_MISC, 0, (double)n * u.hin * u.win * u.cin0 * net->esz * 1.25, c.s);
                if ((rc = launch_upsample2x_bwd(dt, p.out, c.da(u.src0), n, u.hin / 2, u.win / 2, u.cin0, acc0 ? 1 : 0, c.s))) return rc;
            }
            written[u.src0] = 1;
        } else {
            p.out = c.da(u.src0);
            p.residual = written[u.src0] ? c.da(u.src0) : nullptr;
            // If this is the LAST contribution to the gradient of a conv+BN unit's activation and that unit is processed
            // next, its BN-backward reduction can ride in this epilogue (mask, masked store, per-tile partial sums).  Opt-in
            // (fuse_bn_bwd = 1): measured neutral per step.  Per layer (bench.py --per-unit) the BN sweep gets 5-14 us shorter and
            // the dgrad 3-17 us longer: the fused form saves ONE tensor read (the gradient itself) and a launch, but the
            // epilogue reads z / y / earlier contributions 8 bytes per lane (16 scattered 32-byte pieces per instruction) at
            // the tail of every workgroup, which costs about what the separate, fully coalesced sweep cost.  Layers the direct kernel takes
            // are left alone (it has no such epilogue; falling back to the tile kernel cost 55 us each).
            const int pa = u.src0, pu = net->producer[pa];
            if (vs_option("fuse_bn_bwd") && pu >= 0 && pu == ui - 1 && net->first_consumer[pa] == ui && net->units[pu].kind == U_CONV &&
                net->units[pu].bn_idx >= 0 && !(p.Cout & 3) && conv_igemm_variant(dt, p) % 10 != 4) {
                const Unit& q = net->units[pu];
                const int rows_needed = conv_igemm_stat_rows(dt, p);
                // (fixed-point bins for these sums - no finalize launch - were measured in round 3: -0.013 ms per step, and they are not
                // scale-equivariant (gradients of 2 g != 2 x gradients of g bit for bit): not kept)
                if ((size_t)rows_needed * 2 * q.cout * sizeof(float) <= net->bnws_bytes) {
                    const bool recompute = q.relu && q.res < 0;
                    p.bz = c.z(q.out);
                    p.by = (q.relu && !recompute) ? c.a(q.out) : nullptr;
                    p.bmean = c.bnc(q, 2); p.binvstd = c.bnc(q, 3);
                    p.bgamma = c.P(q.bn_idx); p.bbeta = c.P(q.bn_idx + 1);
                    p.bstats_partial = (float*)(c.ws + net->off_bnws);
                    p.brelu = q.relu;
                    net->bwd_stat_rows[pa] = rows_needed;
                }
            }
            prof_set_variant(conv_igemm_variant(dt, p));
            ProfScope prof(PK_CONV_DGRAD, u.kind == U_HEAD ? 2.0 * n * u.hout * u.wout * net->classes * u.k * u.k * u.cin0 : conv_flops(c, u), 0, c.s);
            prof_set_variant(0);
            if ((rc = launch_conv_igemm(dt, p, c.s))) return rc;
            written[u.src0] = 1;
        }
        }
        if (flush) {
            if (do_side) {
                if ((rc = fork_wait(ui))) return rc;
                for (const SideItem& it : pending) {
                    // The stem is the last unit of the backward pass: behind its BatchNorm backward the caller's stream has nothing left
                    // to do but wait for the side stream, which still owes layer1's last weight gradients and their update.  With the fused
                    // optimiser step the stem's weight gradient and update (a group of its own) therefore run on the CALLER's stream, beside
                    // that tail instead of behind it (second slab workspace; profiles/r4_ab_side_stream_experiments.log).
                    if (stem_on_caller && net->units[it.ui].kind == U_STEM) {
                        hipStream_t keep_s = ws_stream;
                        float* keep_w = wgws;
                        ws_stream = c.s;
                        wgws = (float*)(c.ws + net->off_wgws + net->wgws_bytes);
                        rc = side_wgrad(it);
                        ws_stream = keep_s; wgws = keep_w;
                        if (rc) return rc;
                        continue;
                    }
                    if ((rc = side_wgrad(it))) return rc;
                }
            }
            pending.clear();
            prof_set_tag(ui);
        }
    }
    if (!pending.empty()) {   // a range that ends on a unit without weights (the max-pool)
        const int ui = pending.back().ui;
        if (do_main && (rc = fork_mark(ui))) return rc;
        if (do_side) {
            if ((rc = fork_wait(ui))) return rc;
            for (const SideItem& it : pending)
                if ((rc = side_wgrad(it))) return rc;
        }
        pending.clear();
    }
    if (use_side) {  // join: the caller's stream continues only after every weight gradient is in place
        for (int i = 0; i < vs_unet::kSide; ++i) {
            if (!side_used[i]) continue;
            VS_CHECK_HIP(hipEventRecord(net->join_event[i], net->side[i]));
            VS_CHECK_HIP(hipStreamWaitEvent(c.s, net->join_event[i], 0));
        }
    }
    if (opt && role == ROLE_BOTH) net->wset ^= 1;   // split roles: the caller flips once both shares are queued (vs_unet_flip_weight_set)
    return VS_OK;
}

extern "C" int vs_unet_backward_adamw(vs_unet_t* net, const float* x, const float* dlogits, int n, int need_encoder_wgrad,
                                      float* grads, void* workspace, void* stream, const vs_adamw_args* opt) {
    VS_REQUIRE(net && opt && opt->params && opt->exp_avg && opt->exp_avg_sq && opt->step >= 1, "unet_backward_adamw: bad optimiser arguments");
    return unet_backward_range(net, opt->params, x, dlogits, n, need_encoder_wgrad, grads, workspace, stream, 0,
                               (int)net->units.size(), opt);
}

// One stream's share of vs_unet_backward_adamw for the units [unit_lo, unit_hi) (see ROLE_* above): role 1 = the caller's-stream
// kernels (BatchNorm backward, data gradients), role 2 = weight gradients + AdamW + next-forward weight copies, each
// enqueued in order on `stream`; the caller orders role 2 of a range behind role 1 of the same range.  No weight-set flip.
extern "C" int vs_unet_backward_adamw_part(vs_unet_t* net, const float* x, const float* dlogits, int n, int need_encoder_wgrad,
                                           float* grads, void* workspace, void* stream, const vs_adamw_args* opt, int unit_lo,
                                           int unit_hi, int role) {
    VS_REQUIRE(net && opt && opt->params && opt->exp_avg && opt->exp_avg_sq && opt->step >= 1, "unet_backward_adamw_part: bad optimiser arguments");
    VS_REQUIRE(unit_lo >= 0 && unit_lo < unit_hi && unit_hi <= (int)net->units.size(), "unet_backward_adamw_part: bad unit range");
    VS_REQUIRE(role == ROLE_MAIN || role == ROLE_SIDE, "unet_backward_adamw_part: role must be 1 (caller's stream) or 2 (weight gradients)");
    return unet_backward_range(net, opt->params, x, dlogits, n, need_encoder_wgrad, grads, workspace, stream, unit_lo, unit_hi, opt, role);
}

// Data-parallel form of the split roles: the same two shares WITHOUT the optimiser (role 2 = weight gradients only, written to
// `grads`), so the caller can all-reduce a range's gradient slice before vs_unet_adamw_range updates it.
extern "C" int vs_unet_backward_part(vs_unet_t* net, const float* params, const float* x, const float* dlogits, int n,
                                     int need_encoder_wgrad, float* grads, void* workspace, void* stream, int unit_lo,
                                     int unit_hi, int role) {
    VS_REQUIRE(net && unit_lo >= 0 && unit_lo < unit_hi && unit_hi <= (int)net->units.size(), "unet_backward_part: bad unit range");
    VS_REQUIRE(role == ROLE_MAIN || role == ROLE_SIDE, "unet_backward_part: role must be 1 (caller's stream) or 2 (weight gradients)");
    return unet_backward_range(net, params, x, dlogits, n, need_encoder_wgrad, grads, workspace, stream, unit_lo, unit_hi, nullptr, role);
}

// AdamW over the parameters of the units [unit_lo, unit_hi) from `grads` (e.g. an all-reduced slice) plus the weight copies
// of those units for the next forward, in order on `stream`; no weight-set flip (vs_unet_flip_weight_set once every range
// of the step is queued).  The optimiser step of vs_unet_backward_adamw, cut loose from the backward pass.
extern "C" int vs_unet_adamw_range(vs_unet_t* net, int need_encoder_wgrad, const float* grads, void* workspace, void* stream,
                                   const vs_adamw_args* opt, int unit_lo, int unit_hi) {
    VS_REQUIRE(net && grads && workspace && opt && opt->params && opt->exp_avg && opt->exp_avg_sq && opt->step >= 1,
               "unet_adamw_range: bad arguments");
    VS_REQUIRE(unit_lo >= 0 && unit_lo < unit_hi && unit_hi <= (int)net->units.size(), "unet_adamw_range: bad unit range");
    Ctx c{net, (char*)workspace, opt->params, nullptr, (hipStream_t)stream, net->last_n};
    return update_units(c, unit_lo, unit_hi, need_encoder_wgrad != 0, grads, *opt, (hipStream_t)stream);
}

// ---- debug: locate a unit's tensors inside the workspace (tests / diagnostics only) ----------------------
// SyncBatchNorm for data-parallel training: `hook` sums device values in place over the ranks (stream-ordered; see BnSync in common.h),
// `world` ranks contribute equal shares of the global batch.  With it every BatchNorm of a training forward normalises with the
// statistics of the GLOBAL batch (the unit's fixed-point sums are added over the ranks before the normalisation sweep: integer
// sums, so every rank forms the same bits and the result equals a single process running the whole batch), and the backward pass
// sums (sum dy, sum dy xhat) over the ranks for dx.  hook = null: back to per-rank statistics.  Built for bf16 plans whose
// BatchNorms all sit behind bias-free convolutions or the ResNet stem (U-Net / U-Net++ / FPN over the ResNet / ResNeXt encoders -
// BASELINE configs[3]); other plans are refused here.  Replaces nothing in the reference (it is single-GPU: one loader, one
// BatchNorm batch - data/dataloaders.py:42-49); it is what makes N ranks reproduce that single batch.
extern "C" int vs_unet_set_stats_hook(vs_unet_t* net, vs_stats_hook_fn hook, void* user, int world) {
    VS_REQUIRE(net && world >= 1, "vs_unet_set_stats_hook: bad arguments");
    if (hook) {
        VS_REQUIRE(net->dtype == VS_BF16, "vs_unet_set_stats_hook: cross-rank BatchNorm statistics are built for bf16 plans (fixed-point statistics bins)");
        for (const Unit& u : net->units) {
            const bool ok_bn = (u.kind == U_CONV && (u.bn_idx < 0 || (u.bias_idx < 0 && u.off_bins))) || (u.kind == U_STEM && u.off_bins) ||
                               (u.kind != U_CONV && u.kind != U_STEM && u.kind != U_BN && u.kind != U_CONVT && u.kind != U_FPA);
            VS_REQUIRE(ok_bn, "vs_unet_set_stats_hook: this network has BatchNorm layers outside bias-free convolution units (transposed convolutions, "
                              "standalone norms, biased convolutions, the FPA block): SyncBatchNorm is not built for them");
        }
    }
    net->stats_hook = hook; net->stats_user = user; net->stats_world = hook ? world : 1;
    return VS_OK;
}

// Which units' outputs a bf16 training forward at batch n would NOT materialise (normalise-on-load, see nl_consumer): flags[i] = 1
// for such a unit i.  Host logic only - nothing is launched (tests, tools).  Returns the number of units.
extern "C" int vs_unet_nl_plan(vs_unet_t* net, int n, int* flags, int cap) {
    VS_REQUIRE(net && flags && n >= 1 && n <= net->max_batch, "vs_unet_nl_plan: bad arguments");
    Ctx c{net, reinterpret_cast<char*>(uintptr_t(1) << 20), nullptr, nullptr, nullptr, n};   // (addresses are never dereferenced)
    std::vector<char> act(net->acts.size(), 0);
    for (int i = 0; i < (int)net->units.size(); ++i) {
        const Unit& u = net->units[i];
        if (i < cap) flags[i] = 0;
        if (u.kind != U_CONV || net->dtype != VS_BF16 || u.bn_idx < 0 || u.gn_idx >= 0 || u.bias_idx >= 0) continue;
        if (!vs_option("stats_bins") || !net->bins_bytes) continue;
        ConvParams p = conv_params(c, u);
        if (!conv_igemm_bins_ok(net->dtype, p) || nl_consumer(c, i) < 0) continue;
        act[u.out] = 1;
        if (i < cap) flags[i] = 1;
    }
    return (int)net->units.size();
}

extern "C" int vs_unet_num_units(const vs_unet_t* net) { return (int)net->units.size(); }
extern "C" int vs_unet_debug_unit(const vs_unet_t* net, int unit, char* wname, int name_len, int* c, int* h, int* w,
                                  size_t* off_a, size_t* off_z, size_t* off_da, size_t* off_dz) {
    VS_REQUIRE(net && unit >= 0 && unit < (int)net->units.size(), "debug_unit: bad index");
    const Unit& u = net->units[unit];
    const char* nm = u.w_idx >= 0 ? net->layout.tensors[u.w_idx].name.c_str() : u.kind == U_BN ? net->layout.tensors[u.bn_idx].name.c_str() : u.kind == U_DROPADD ? "drop_connect+add" : u.kind == U_RADIXSUM ? "radix-weighted sum" : u.kind == U_FOLD2 ? "radix fold" : u.kind == U_RSOFTMAX ? "radix softmax" : u.kind == U_AVGPOOL ? "avgpool k" : (u.kind == U_CONCAT ? "concat" : (u.kind == U_ADD ? "add" : (u.kind == U_UPADD ? "upsample+add" : (u.kind == U_BILINEAR ? "bilinear"
                         : (u.kind == U_DROPOUT ? "dropout2d" : (u.kind == U_GAP ? "avgpool" : (u.kind == U_BCAST ? "broadcast" : (u.kind == U_DROPOUT_E ? "dropout"
                         : (u.kind == U_PAB ? "pab attention" : (u.kind == U_CGATE ? "channel gate" : (u.kind == U_SIGMOID ? "sigmoid" : "maxpool")))))))))));
    strncpy(wname, nm, name_len - 1); wname[name_len - 1] = 0;
    if (u.out < 0) { *c = *h = *w = 0; *off_a = *off_z = *off_da = *off_dz = 0; return VS_OK; }
    const Act& a = net->acts[u.out];
    *c = a.c; *h = a.h; *w = a.w; *off_a = a.off_a; *off_z = a.off_z; *off_da = a.off_da; *off_dz = a.off_dz;
    return VS_OK;
}
