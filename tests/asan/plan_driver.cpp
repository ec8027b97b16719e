// Host-side sanitizer run of libvolseg_hip's planning code (tests/asan/Makefile): every member of the model matrix the engine
// builds x class counts x slice sizes x batch sizes - plan construction, tensor tables, workspace layout, unit / parameter
// maps, the error paths - under AddressSanitizer + UBSan.  No GPU is touched: nothing is launched.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "volseg_hip.h"

#define CHECK(cond)                                                       \
    do {                                                                  \
        if (!(cond)) {                                                    \
            fprintf(stderr, "%s:%d: CHECK failed: %s (%s)\n", __FILE__, __LINE__, #cond, vs_last_error()); \
            return 1;                                                     \
        }                                                                 \
    } while (0)

int main() {
    const int encoders[] = {18, 34, 50, 51, 1018, 1034, 1050, 1051, 2018, 2034, 2050, 2051, 3018, 3034, 3050, 3051, 4018, 4034, 4050, 5018, 5034, 5050, 6018, 6034, 6050, 6051, 7018, 7034, 7050, 103, 104, 1103, 1104, 3103, 3104, 4103, 4104, 5103, 5104, 6103, 6104, 7103, 7104, 150, 201, 1150, 2150, 3150, 6150, 6201};     // topology * 1000 + depth
    long plans = 0;
    for (int enc : encoders) {
        for (int classes : {1, 2, 4, 16}) {
            const int nt = vs_unet_num_tensors_ex(classes, enc);
            CHECK(nt > 100);
            int64_t params = 0, bn = 0;
            std::string prev;
            for (int i = 0; i < nt; ++i) {
                char name[128];
                int64_t shape[4], off;
                int ndim, kind;
                CHECK(vs_unet_tensor_info_ex(classes, enc, i, name, sizeof(name), shape, &ndim, &kind, &off) == VS_OK);
                int64_t numel = 1;
                for (int d = 0; d < ndim; ++d) numel *= shape[d];
                CHECK(numel > 0 && ndim >= 1 && ndim <= 4 && kind >= 0 && kind <= 5 && strlen(name) > 0);
                if (kind <= 3) { CHECK(off == params); params += numel; } else { CHECK(off == bn); bn += numel; }   // dense, in order
            }
            CHECK(params == vs_unet_param_elems_ex(classes, enc) && bn == vs_unet_bnstate_elems_ex(classes, enc));
            char tiny[4];
            int64_t shape[4], off; int ndim, kind;
            CHECK(vs_unet_tensor_info_ex(classes, enc, 0, tiny, sizeof(tiny), shape, &ndim, &kind, &off) == VS_OK && strlen(tiny) == 3);   // truncating copy
            CHECK(vs_unet_tensor_info_ex(classes, enc, nt, tiny, sizeof(tiny), shape, &ndim, &kind, &off) != VS_OK);
        }
        for (int dtype : {VS_F32, VS_BF16})
            for (int hw : {32, 64, 96, 256, 512})
                for (int batch : {1, 3, 32}) {
                    vs_unet_t* net = nullptr;
                    CHECK(vs_unet_create_ex(&net, dtype, 2 + (hw & 2), batch, hw, hw == 96 ? 64 : hw, enc) == VS_OK && net);
                    const size_t we = vs_unet_workspace_bytes(net, 0), wt = vs_unet_workspace_bytes(net, 1);
                    CHECK(we > 0 && wt > we);
                    const int nu = vs_unet_num_units(net);
                    CHECK(nu > 20);
                    const int64_t n_params = vs_unet_param_elems_ex(2 + (hw & 2), enc);
                    for (int u = 0; u <= nu; ++u) {   // (a block's 1x1 shortcut runs before its last convolution but is registered after it:
                        const int64_t o = vs_unet_unit_param_offset(net, u);   //  offsets are not monotonic in the unit index)
                        CHECK(o >= 0 && o <= n_params);
                    }
                    CHECK(vs_unet_unit_param_offset(net, nu) == n_params && vs_unet_unit_param_offset(net, 0) == 0);
                    for (int u = 0; u < nu; ++u) {
                        char nm[96]; int c, h, w; size_t oa, oz, oda, odz;
                        CHECK(vs_unet_debug_unit(net, u, nm, sizeof(nm), &c, &h, &w, &oa, &oz, &oda, &odz) == VS_OK);
                        const size_t bytes = (size_t)batch * c * h * w * (dtype == VS_BF16 ? 2 : 4);
                        CHECK(oa + bytes <= wt && oda + bytes <= wt && oz + (oz ? bytes : 0) <= wt && odz + (odz ? bytes : 0) <= wt);
                    }
                    CHECK(vs_unet_weight_set(net) == 0 && vs_unet_flip_weight_set(net) == VS_OK && vs_unet_weight_set(net) == 1);
                    vs_unet_destroy(net);
                    ++plans;
                }
    }
    // error paths return codes and messages, never touch memory they should not
    vs_unet_t* net = nullptr;
    CHECK(vs_unet_create_ex(&net, VS_BF16, 2, 4, 100, 64, 34) == VS_ERR_INVALID && strstr(vs_last_error(), "multiples of 32"));
    CHECK(vs_unet_create_ex(&net, VS_BF16, 2, 4, 64, 64, 33) == VS_ERR_INVALID && vs_unet_create_ex(&net, 7, 2, 4, 64, 64, 34) == VS_ERR_INVALID);
    CHECK(vs_unet_create_ex(&net, VS_BF16, 0, 4, 64, 64, 34) == VS_ERR_INVALID && vs_unet_create_ex(&net, VS_BF16, 2, 4, 64, 64, 8034) == VS_ERR_INVALID);
    CHECK(vs_unet_create_ex(&net, VS_BF16, 2, 4, 64, 64, 2104) == VS_ERR_INVALID && strstr(vs_last_error(), "EfficientNet"));     // not under Linknet
    CHECK(vs_unet_create_ex(&net, VS_BF16, 2, 4, 64, 64, 4150) == VS_ERR_INVALID && strstr(vs_last_error(), "ResNeSt"));          // not under DeepLabV3+
    CHECK(vs_unet_num_tensors_ex(99, 34) < 0 && vs_unet_param_elems_ex(2, 35) < 0);
    CHECK(vs_set_option("no_such_option", 1) == VS_ERR_INVALID && vs_set_option("fork_every", 2) == VS_OK && vs_get_option("fork_every") == 2);
    vs_conv_desc d{};
    d.dtype = VS_BF16; d.n = 2; d.hin = 64; d.win = 64; d.c0 = 64; d.cout = 64; d.kh = d.kw = 3; d.stride = 1; d.pad = 1;
    CHECK(vs_conv2d_wgrad_workspace(&d) > 0);
    CHECK(vs_bn_workspace(1000, 64) > 0 && vs_dice_workspace(4) > 0 && vs_seg_loss_workspace(4) > 0 && vs_mean_iou_workspace(3, 4) > 0);
    printf("plan_driver: %ld plans built and torn down under ASan + UBSan, no findings\n", plans);
    return 0;
}
