// Shared device/host helpers for the timediff_hip kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/timediff_hip.h"

typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f32x4;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;

#define TD_WAVE 64

// ---- error plumbing --------------------------------------------------------------------------
void td_set_error(const char *fmt, ...);

#define TD_REQUIRE(cond, ...)              \
    do {                                   \
        if (!(cond)) {                     \
            td_set_error(__VA_ARGS__);     \
            return TD_ERR_INVALID;         \
        }                                  \
    } while (0)

#define TD_HIP(call)                                                                              \
    do {                                                                                          \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess) {                                                                   \
            td_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            return TD_ERR_HIP;                                                                    \
        }                                                                                         \
    } while (0)

#define TD_LAUNCH_CHECK() TD_HIP(hipGetLastError())

static inline int td_cdiv(int a, int b) { return (a + b - 1) / b; }

// ---- device math -----------------------------------------------------------------------------
#define TD_LOG_1E30 (-69.07755279f)  // log(clamp(one_hot, 1e-30)), reference mixed_diffusion.py:50

// factors folded into the packed recurrent weights (td_pack_weights_kernel) so that the layer kernel's gate math
// starts at exp2: -log2(e) for the sigmoid gates, 2 log2(e) for the tanh gate
#define TD_GATE_SCALE_SIGMOID (-1.4426950408889634f)
#define TD_GATE_SCALE_TANH (2.8853900817779268f)

// v_exp_f32 / v_rcp_f32 (1 ulp) forms: these run 5x per hidden unit per time step
__device__ __forceinline__ float td_sigmoid(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
__device__ __forceinline__ float td_tanh(float x) {
    // 1 - 2/(e^{2x}+1): saturates cleanly to +-1, abs error ~1e-7
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(2.8853900817779268f * x) + 1.0f);
}
// reference mixed_diffusion.py:20-22, same operation order in fp32
__device__ __forceinline__ float td_log_add_exp(float a, float b) {
    float m = fmaxf(a, b);
    return m + logf(expf(a - m) + expf(b - m));
}

