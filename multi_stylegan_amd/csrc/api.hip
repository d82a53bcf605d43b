// Library identity and error strings of the C ABI (include/msg_hip.h).
#include "msg_common.h"

// 3: deterministic reductions (workspace arguments of msg_conv2d_wgrad, msg_bias_act_backward)
// 4: msg_scale_reduce_channels / msg_scale_bias_act removed, msg_rgb_skip_merge(+_backward) and the storage code MSG_F32_SPLIT
//    added, msg_affine_warp's backward workspace grew to B*C*H*W + 1 words (the poison word)
// 5: round 5 entries (see include/msg_hip.h: "ABI 5")
extern "C" int msg_abi_version(void) { return MSG_ABI_VERSION; }
extern "C" const char* msg_build_arch(void) { return "gfx950"; }
extern "C" const char* msg_strerror(int code) {
    switch (code) {
        case MSG_OK: return "ok";
        case MSG_EINVAL: return "invalid argument (null pointer, non-positive extent or inconsistent sizes)";
        case MSG_EUNSUPPORTED: return "configuration not supported by the gfx950 kernels";
        case MSG_ELAUNCH: return "kernel launch failed (hipGetLastError != hipSuccess)";
        default: return "unknown msg_hip error code";
    }
}
