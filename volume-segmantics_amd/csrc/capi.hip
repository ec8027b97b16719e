// C-ABI glue: error reporting and the single-operator entry points declared in include/volseg_hip.h.
#include <stdarg.h>

#include "common.h"

static thread_local char g_err[512] = "";

void vs_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* vs_last_error(void) { return g_err; }
extern "C" int vs_version(void) { return 100; }

static int desc_to_params(const vs_conv_desc* d, ConvParams& p) {
    VS_REQUIRE(d, "conv: null descriptor");
    p = ConvParams{};
    p.C0 = d->c0; p.C1 = d->c1; p.up0 = d->up0;
    p.N = d->n; p.Hin = d->hin; p.Win = d->win;
    p.stride = d->stride; p.pad = d->pad; p.KH = d->kh; p.KW = d->kw;
    VS_REQUIRE(d->stride >= 1 && d->kh >= 1 && d->kw >= 1 && d->dilation >= 0, "conv: bad geometry");
    const int dil = d->dilation > 1 ? d->dilation : 1;
    p.dil = dil;
    p.Hout = (d->hin + 2 * d->pad - (d->kh - 1) * dil - 1) / d->stride + 1;
    p.Wout = (d->win + 2 * d->pad - (d->kw - 1) * dil - 1) / d->stride + 1;
    p.Cout = d->cout; p.relu = d->relu; p.out_f32 = d->out_f32; p.split_c = d->split_c;
    VS_REQUIRE(d->groups >= 0, "conv: bad group count");
    if (d->groups > 1) {   // grouped: runs on 32-channel super-groups (weights from vs_weights_prepare_grouped)
        VS_REQUIRE(d->c1 == 0 && d->c0 == d->cout && d->c0 % d->groups == 0 && 32 % (d->c0 / d->groups) == 0 && d->c0 % 32 == 0 &&
                   d->c0 / d->groups >= 4, "conv: grouped convolutions need c0 == cout, 4 / 8 / 16 / 32 channels per group");
        p.gc = 32;
    }
    return VS_OK;
}

extern "C" int vs_conv2d_fwd(const vs_conv_desc* d, const void* src0, const void* src1, const void* w,
                             const float* scale, const float* shift, const void* residual, void* y, void* y1,
                             void* stream) {
    ConvParams p;
    int rc = desc_to_params(d, p);
    if (rc) return rc;
    VS_REQUIRE((d->c1 == 0) == (src1 == nullptr), "conv: src1 / c1 mismatch");
    VS_REQUIRE((d->split_c > 0) == (y1 != nullptr), "conv: y1 / split_c mismatch");
    p.src0 = src0; p.src1 = src1; p.w = w; p.scale = scale; p.shift = shift; p.residual = residual;
    p.out = y; p.out1 = y1;
    return launch_conv_igemm(d->dtype, p, (hipStream_t)stream);
}

// which kernel instantiation vs_conv2d_fwd picks for this descriptor (with an affine epilogue where the descriptor allows one):
// cout tile * 1000 + pixel tiles per wave * 100 + taps * 10 + kind (1 = stride-1 tile kernel, 2 = stride 2, 4 = direct shallow-layer
// kernel, 6 = LDS-DMA ring, 7 = persistent LDS-DMA ring, 8 = 8-wave 256-pixel tiles); negative = error.  For tests and tools.
extern "C" int vs_conv2d_variant(const vs_conv_desc* d) {
    ConvParams p;
    int rc = desc_to_params(d, p);
    if (rc) return rc;
    p.src0 = (const void*)16; p.src1 = d->c1 ? (const void*)16 : nullptr; p.w = (const void*)16; p.out = (void*)16;
    p.out1 = d->split_c > 0 ? (void*)16 : nullptr;
    return conv_igemm_variant(d->dtype, p);
}

// the training forms of the launch (statistics epilogue, pooled / masked data gradients, normalise on load): ConvParams as the
// network plan fills it, from the flat C structure
static int train_to_params(const vs_conv_desc* d, const vs_conv_train* t, ConvParams& p) {
    int rc = desc_to_params(d, p);
    if (rc) return rc;
    VS_REQUIRE(t, "conv_train: null training block");
    VS_REQUIRE(!(t->stats_bins && t->stats_partial), "conv_train: statistics go to bins OR partial rows");
    VS_REQUIRE(!t->stats_bins || (t->stats_nb >= 1 && (t->stats_nb & (t->stats_nb - 1)) == 0), "conv_train: stats_nb must be a power of two");
    p.stats_bins = (unsigned long long*)t->stats_bins; p.stats_nb = t->stats_nb; p.stats_partial = t->stats_partial;
    p.pool0 = t->pool0;
    p.bz = t->bz; p.by = t->by; p.bmean = t->bmean; p.binvstd = t->binvstd; p.bgamma = t->bgamma; p.bbeta = t->bbeta;
    p.bstats_partial = t->bstats_partial; p.brelu = t->brelu;
    if (t->nl_bins) {
        VS_REQUIRE(t->nl_nb >= 1 && t->nl_rows >= 1 && t->nl_mean && t->nl_invstd && t->nl_gamma && t->nl_beta && t->nl_y,
                   "conv_train: normalise-on-load needs its bins, row count, affine parameters and outputs");
        p.nl_bins = (const unsigned long long*)t->nl_bins; p.nl_nb = t->nl_nb; p.nl_rows = t->nl_rows; p.nl_eps = t->nl_eps; p.nl_mom = t->nl_mom;
        p.nl_mean = t->nl_mean; p.nl_invstd = t->nl_invstd; p.nl_rm = t->nl_rm; p.nl_rv = t->nl_rv;
        p.nl_gamma = t->nl_gamma; p.nl_beta = t->nl_beta; p.nl_y = t->nl_y;
    }
    return VS_OK;
}

extern "C" int vs_conv2d_train(const vs_conv_desc* d, const void* src0, const void* src1, const void* w, const void* residual, void* y,
                               void* y1, const vs_conv_train* t, void* stream) {
    ConvParams p;
    int rc = train_to_params(d, t, p);
    if (rc) return rc;
    VS_REQUIRE((d->c1 == 0) == (src1 == nullptr), "conv: src1 / c1 mismatch");
    VS_REQUIRE((d->split_c > 0) == (y1 != nullptr), "conv: y1 / split_c mismatch");
    p.src0 = src0; p.src1 = src1; p.w = w; p.residual = residual; p.out = y; p.out1 = y1;
    if (p.pool0) VS_REQUIRE(conv_igemm_can_pool(p), "conv_train: no pooled epilogue for this geometry");
    if (p.nl_bins) VS_REQUIRE(conv_igemm_nl_ok(d->dtype, p), "conv_train: no normalise-on-load form for this layer");
    if (p.stats_bins) VS_REQUIRE(conv_igemm_bins_ok(d->dtype, p), "conv_train: this layer's kernel has no statistics bins");
    return launch_conv_igemm(d->dtype, p, (hipStream_t)stream);
}

static void fake_pointers(const vs_conv_desc* d, ConvParams& p) {
    p.src0 = (const void*)16; p.src1 = d->c1 ? (const void*)16 : nullptr; p.w = (const void*)16; p.out = (void*)16;
    p.out1 = d->split_c > 0 ? (void*)16 : nullptr;
}
extern "C" int vs_conv2d_train_variant(const vs_conv_desc* d, const vs_conv_train* t) {
    ConvParams p;
    int rc = train_to_params(d, t, p);
    if (rc) return rc;
    fake_pointers(d, p);
    return conv_igemm_variant(d->dtype, p);
}
extern "C" int vs_conv2d_stat_rows(const vs_conv_desc* d, const vs_conv_train* t) {
    ConvParams p;
    int rc = train_to_params(d, t, p);
    if (rc) return rc;
    fake_pointers(d, p);
    return conv_igemm_stat_rows(d->dtype, p);
}
extern "C" double vs_stat_scale(int which) { return which == 0 ? kStatScale1 : kStatScale2; }

static int pair_params(const vs_conv_desc* d1, const vs_conv_desc* d2, ConvParams& p, ConvParams& q) {
    int rc = desc_to_params(d1, p);
    if (rc) return rc;
    if ((rc = desc_to_params(d2, q))) return rc;
    VS_REQUIRE(d1->dtype == d2->dtype && d1->c1 == 0 && d2->c1 == 0 && d1->split_c == 0 && d2->split_c == 0, "conv_pair: plain single-source layers of one dtype");
    return VS_OK;
}
extern "C" int vs_conv2d_pair_ok(const vs_conv_desc* d1, const vs_conv_desc* d2) {
    ConvParams p, q;
    if (pair_params(d1, d2, p, q)) return 0;
    fake_pointers(d1, p); fake_pointers(d2, q);
    return conv_pair_ok(d1->dtype, p, q) ? 1 : 0;
}
extern "C" int vs_conv2d_pair_fwd(const vs_conv_desc* d1, const vs_conv_desc* d2, const void* src0, const void* w1, const float* scale1,
                                  const float* shift1, const void* w2, const float* scale2, const float* shift2, void* y, void* stream) {
    ConvParams p, q;
    int rc = pair_params(d1, d2, p, q);
    if (rc) return rc;
    p.src0 = src0; p.w = w1; p.scale = scale1; p.shift = shift1;
    q.src0 = (const void*)16; q.w = w2; q.scale = scale2; q.shift = shift2; q.out = y;      // (q's source is the tensor that is never written)
    return launch_conv_pair(d1->dtype, p, q, (hipStream_t)stream);
}

static int desc_to_wgrad(const vs_conv_desc* d, WgradParams& p) {
    VS_REQUIRE(d, "conv: null descriptor");
    p = WgradParams{};
    p.C0 = d->c0; p.C1 = d->c1; p.up0 = d->up0; p.N = d->n; p.Hin = d->hin; p.Win = d->win;
    p.stride = d->stride; p.pad = d->pad; p.KH = d->kh; p.KW = d->kw;
    const int dil = d->dilation > 1 ? d->dilation : 1;
    p.dil = dil;
    p.Hout = (d->hin + 2 * d->pad - (d->kh - 1) * dil - 1) / d->stride + 1;
    p.Wout = (d->win + 2 * d->pad - (d->kw - 1) * dil - 1) / d->stride + 1;
    p.Cout = d->cout;
    if (d->groups > 1) {
        VS_REQUIRE(d->c1 == 0 && d->c0 == d->cout && d->c0 % d->groups == 0 && 32 % (d->c0 / d->groups) == 0 && d->c0 % 32 == 0 &&
                   d->c0 / d->groups >= 4, "conv: grouped convolutions need c0 == cout, 4 / 8 / 16 / 32 channels per group");
        p.cg = d->c0 / d->groups;
    }
    return VS_OK;
}

extern "C" size_t vs_conv2d_wgrad_workspace(const vs_conv_desc* d) {
    WgradParams p;
    if (desc_to_wgrad(d, p)) return 0;
    return wgrad_workspace_bytes(d->dtype, p);
}

extern "C" int vs_conv2d_wgrad(const vs_conv_desc* d, const void* src0, const void* src1, const void* dy, float* dw,
                               void* workspace, size_t workspace_bytes, void* stream) {
    WgradParams p;
    int rc = desc_to_wgrad(d, p);
    if (rc) return rc;
    VS_REQUIRE(src0 && dy && dw, "conv_wgrad: null pointer");
    p.src0 = src0; p.src1 = src1; p.dy = dy; p.dw = dw;
    p.partials = (float*)workspace; p.partial_bytes = workspace_bytes;
    return launch_conv_wgrad(d->dtype, p, (hipStream_t)stream);
}

/* The segmentation head's backward straight from dLoss / dlogits as autograd hands it over (fp32 NCHW planes of `classes` channels, no
 * 16-channel NHWC copy in between).  vs_head_dgrad_planes: the gradient of the head's 16-bit NHWC input ([n][h][w][c], c <= 16) from the
 * forward's weight copy w ([classes][9][c]); returns VS_ERR_UNSUPPORTED outside 9 * classes <= 64.  vs_head_wgrad_planes: the weight gradient
 * dw fp32 [classes][9][16] of a 16-channel input x; VS_ERR_UNSUPPORTED where the row-streaming kernel does not apply (classes > 7, widths other
 * than 128 / 256 / 512).  Both are what vs_unet_backward* runs for the head where they apply. */
extern "C" int vs_head_dgrad_planes(int dtype, const float* dlogits, const void* w, void* dx, int n, int classes, int h, int wd, int c, void* stream) {
    VS_REQUIRE(dlogits && w && dx, "head_dgrad_planes: null pointer");
    if (!head_dgrad_planes_ok(dtype, classes, h, wd, c)) { vs_set_error("head_dgrad_planes: unsupported shape"); return VS_ERR_UNSUPPORTED; }
    return launch_head_dgrad_planes(dtype, dlogits, w, dx, n, classes, h, wd, c, (hipStream_t)stream);
}
static WgradParams head_wgrad_params(int n, int classes, int h, int wd) {
    WgradParams p{};
    p.C0 = 16; p.N = n; p.Hin = p.Hout = h; p.Win = p.Wout = wd; p.stride = 1; p.pad = 1; p.KH = p.KW = 3; p.Cout = 16;
    p.cout_live = classes; p.dy_planes = classes;
    return p;
}
extern "C" size_t vs_head_wgrad_planes_workspace(int dtype, int n, int classes, int h, int wd) {
    const WgradParams p = head_wgrad_params(n, classes, h, wd);
    return conv_wgrad_takes_planes(dtype, p) ? wgrad_workspace_bytes(dtype, p) : 0;
}
extern "C" int vs_head_wgrad_planes(int dtype, const void* x, const float* dlogits, float* dw, void* workspace, size_t workspace_bytes, int n, int classes,
                                    int h, int wd, void* stream) {
    VS_REQUIRE(x && dlogits && dw, "head_wgrad_planes: null pointer");
    WgradParams p = head_wgrad_params(n, classes, h, wd);
    if (!conv_wgrad_takes_planes(dtype, p)) { vs_set_error("head_wgrad_planes: unsupported shape"); return VS_ERR_UNSUPPORTED; }
    p.src0 = x; p.dy = dlogits; p.dw = dw; p.partials = (float*)workspace; p.partial_bytes = workspace_bytes;
    return launch_conv_wgrad(dtype, p, (hipStream_t)stream);
}

int launch_weight_prepare(int dtype, const float* w, void* wc, void* wt, int cout, int taps, int cin, int cout_pad, hipStream_t s);
/* fp32 [cout][taps][cin] -> dtype copy (wc, may be null) and flipped/transposed dgrad copy [cin][taps][cout] (wt) */
extern "C" int vs_weights_prepare(int dtype, const float* w, void* wc, void* wt, int cout, int taps, int cin, void* stream) {
    VS_REQUIRE(w && (wc || wt), "weights_prepare: null pointer");
    return launch_weight_prepare(dtype, w, wc, wt, cout, taps, cin, cout, (hipStream_t)stream);
}
/* grouped convolution with cg = cin / groups channels per group (cin == cout): fp32 [cout][taps][cg] -> wc [cout][taps][32]
 * (the group's block inside its 32-channel super-group, zeros elsewhere) and wt [cin][taps reversed][32] for the data gradient */
extern "C" int vs_weights_prepare_grouped(int dtype, const float* w, void* wc, void* wt, int cout, int taps, int cg, void* stream) {
    VS_REQUIRE(w && (wc || wt), "weights_prepare_grouped: null pointer");
    return launch_weight_prepare_grouped(dtype, w, wc, wt, cout, taps, cg, (hipStream_t)stream);
}

/* nn.ConvTranspose2d(kernel 4, stride 2, padding 1) as a 3x3 convolution onto 4 * cout channels + vs_depth_to_space2: w fp32
 * [cin][cout][4][4] (torch's layout) -> wc [4 * cout][9][cin] (may be NULL) and the data-gradient copy wt [cin][9][4 * cout]
 * (may be NULL); vs_convt_wgrad_gather maps the 3x3 form's weight gradient (vs_conv2d_wgrad, [4 * cout][9][cin] fp32) back. */
extern "C" int vs_convt_weights_prepare(int dtype, const float* w, void* wc, void* wt, int cin, int cout, void* stream) {
    VS_REQUIRE(w && (wc || wt) && cin > 0 && cout > 0, "convt_weights_prepare: bad arguments");
    return launch_convt_weight_prepare(dtype, w, wc, wt, cin, cout, (hipStream_t)stream);
}
extern "C" int vs_convt_wgrad_gather(const float* dense, float* dw, int cin, int cout, void* stream) {
    VS_REQUIRE(dense && dw && cin > 0 && cout > 0, "convt_wgrad_gather: bad arguments");
    return launch_convt_wgrad_gather(dense, dw, cin, cout, (hipStream_t)stream);
}
