// Counter-based noise of the diffusion kernels (Philox4x32-10).  No counterpart in the reference (it draws from torch's
// global generator); parity runs feed external noise.  The draw layout is shared with oracle/etdiff_oracle.c.
#pragma once
#include "common.h"

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
        // angle 2 pi u2 with u2 = k / 2^24: sincospi reduces the argument exactly (sinf / cosf of a rounded 2 pi u2 carry
        // a Payne-Hanek path the compiler cannot drop: 40 % of the reverse-update kernel's instructions)
        float sn, cs;
        sincospif(2.0f * u2, &sn, &cs);
        z[2 * p] = rad * cs;
        z[2 * p + 1] = rad * sn;
    }
}
// Uniform block `block` (counter word 3 = 0x10000 + block) yields the flat class slots 5 block .. 5 block + 4: the 128
// output bits are cut into five 24-bit fields (a uniform here carries 24 bits anyway), most significant first, so the
// 10 uniforms of an eICU-shaped position cost two Philox calls instead of three.
#define TD_UNI_PER_BLOCK 5
__device__ __forceinline__ void td_uniform5(uint64_t seed, uint32_t patient, uint32_t l, uint32_t tag, uint32_t block,
                                            float u[5]) {
    Philox4 r = td_philox(patient, l, tag, 0x10000u + block, (uint32_t)seed, (uint32_t)(seed >> 32));
    const float s = 1.0f / 16777216.0f;
    u[0] = (float)(r.v[0] >> 8) * s;
    u[1] = (float)(((r.v[0] & 0xFFu) << 16) | (r.v[1] >> 16)) * s;
    u[2] = (float)(((r.v[1] & 0xFFFFu) << 8) | (r.v[2] >> 24)) * s;
    u[3] = (float)(r.v[2] & 0xFFFFFFu) * s;
    u[4] = (float)(r.v[3] >> 8) * s;
}
