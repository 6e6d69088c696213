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

// ---- Philox4x32-10, counter layout shared with oracle/etdiff_oracle.c --------------------------
struct Philox4 {
    uint32_t v[4];
};
__host__ __device__ __forceinline__ Philox4 td_philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                                      uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    Philox4 o;
    o.v[0] = c0; o.v[1] = c1; o.v[2] = c2; o.v[3] = c3;
    return o;
}
__host__ __device__ __forceinline__ float td_u01(uint32_t x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }
__host__ __device__ __forceinline__ float td_u01_open(uint32_t x) {
    return ((float)(x >> 8) + 1.0f) * (1.0f / 16777216.0f);
}
// channel ch's standard normal for (patient, l, tag): block ch/4, element ch%4 (two Box-Muller pairs)
__device__ __forceinline__ void td_normal4(uint64_t seed, uint32_t patient, uint32_t l, uint32_t tag, uint32_t block,
                                           float z[4]) {
    Philox4 r = td_philox(patient, l, tag, block, (uint32_t)seed, (uint32_t)(seed >> 32));
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        float u1 = td_u01_open(r.v[2 * p]), u2 = td_u01(r.v[2 * p + 1]);
        float rad = sqrtf(-2.0f * logf(u1));
        float ang = 6.283185307179586f * u2;
        z[2 * p] = rad * cosf(ang);
        z[2 * p + 1] = rad * sinf(ang);
    }
}
__device__ __forceinline__ void td_uniform4(uint64_t seed, uint32_t patient, uint32_t l, uint32_t tag, uint32_t block,
                                            float u[4]) {
    Philox4 r = td_philox(patient, l, tag, 0x10000u + block, (uint32_t)seed, (uint32_t)(seed >> 32));
#pragma unroll
    for (int q = 0; q < 4; ++q) u[q] = td_u01(r.v[q]);
}
