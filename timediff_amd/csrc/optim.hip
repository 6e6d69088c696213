// Fused optimizer pass over the flat parameter buffer (et_diff.py:239-254): global L2 grad-norm,
// clip_grad_norm_(1.0), Adam (torch.optim.Adam defaults: no weight decay, no amsgrad) and the EMA update
// in two launches instead of ~10 foreach kernels per tensor list.
#include "common.h"

#define TD_NORM_BLOCKS 1024

__global__ __launch_bounds__(256) void td_sqnorm_kernel(const float *__restrict__ g, size_t n, double *partial) {
    double s = 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) s += (double)g[i] * g[i];
    __shared__ double red[256];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

__global__ __launch_bounds__(256) void td_sqnorm_finish_kernel(const double *partial, int n, float *norm) {
    __shared__ double red[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) s += partial[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) norm[0] = (float)sqrt(red[0]);
}

extern "C" size_t td_grad_norm_workspace_bytes(void) { return TD_NORM_BLOCKS * sizeof(double); }

extern "C" int td_grad_norm(const float *grad, size_t n, float *norm, void *workspace, void *stream) {
    TD_REQUIRE(grad && norm && workspace && n > 0, "bad argument");
    size_t b = (n + 255) / 256;
    const int blocks = (int)(b > TD_NORM_BLOCKS ? TD_NORM_BLOCKS : b);
    hipLaunchKernelGGL(td_sqnorm_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, grad, n, (double *)workspace);
    TD_LAUNCH_CHECK();
    hipLaunchKernelGGL(td_sqnorm_finish_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const double *)workspace, blocks, norm);
    TD_LAUNCH_CHECK();
    return TD_OK;
}

struct TdAdamParams {
    float *p, *m, *v, *ema;
    const float *g, *norm;
    size_t n;
    float max_norm, step_size, inv_sqrt_bc2, beta1, beta2, eps, ema_decay;
    int ema_mode;  // 0: leave the EMA alone, 1: copy the new weights, 2: ema = decay * ema + (1 - decay) * w
    const td_step_record *rec;   // non-NULL: step_size, inv_sqrt_bc2, ema_decay and ema_mode come from this device record
};

__global__ __launch_bounds__(256) void td_clip_adam_ema_kernel(TdAdamParams a) {
    // clip_grad_norm_: coef = min(1, max_norm / (norm + 1e-6))
    const float coef = a.max_norm > 0.f ? fminf(1.0f, a.max_norm / (a.norm[0] + 1e-6f)) : 1.0f;
    if (a.rec) {
        a.step_size = a.rec->step_size; a.inv_sqrt_bc2 = a.rec->inv_sqrt_bc2;
        a.ema_decay = a.rec->ema_decay; a.ema_mode = a.rec->ema_mode;
    }
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < a.n; i += (size_t)gridDim.x * 256) {
        const float g = a.g[i] * coef;
        const float m = a.beta1 * a.m[i] + (1.0f - a.beta1) * g;
        const float v = a.beta2 * a.v[i] + (1.0f - a.beta2) * g * g;
        a.m[i] = m;
        a.v[i] = v;
        const float w = a.p[i] - a.step_size * m / (sqrtf(v) * a.inv_sqrt_bc2 + a.eps);
        a.p[i] = w;
        if (a.ema_mode == 1) a.ema[i] = w;
        else if (a.ema_mode == 2) a.ema[i] = a.ema_decay * a.ema[i] + (1.0f - a.ema_decay) * w;
    }
}

extern "C" int td_clip_adam_ema(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, float *ema, size_t n,
                                const float *norm, float max_norm, float lr, float beta1, float beta2, float eps,
                                int32_t step, int32_t ema_mode, float ema_decay, void *stream) {
    TD_REQUIRE(param && grad && exp_avg && exp_avg_sq && n > 0 && step >= 1, "bad argument");
    TD_REQUIRE(max_norm <= 0.f || norm, "norm pointer required when clipping");
    TD_REQUIRE(ema_mode == 0 || ema, "ema pointer required");
    TdAdamParams a;
    a.p = param; a.g = grad; a.m = exp_avg; a.v = exp_avg_sq; a.ema = ema; a.norm = norm; a.n = n;
    a.max_norm = max_norm; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.ema_decay = ema_decay; a.ema_mode = ema_mode;
    a.step_size = (float)((double)lr / (1.0 - pow((double)beta1, (double)step)));
    a.inv_sqrt_bc2 = (float)(1.0 / sqrt(1.0 - pow((double)beta2, (double)step)));
    a.rec = nullptr;
    size_t b = (n + 255) / 256;
    const int blocks = (int)(b > 4096 ? 4096 : b);
    hipLaunchKernelGGL(td_clip_adam_ema_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
    TD_LAUNCH_CHECK();
    return TD_OK;
}

__global__ void td_step_record_kernel(td_step_record *dst, td_step_record v) { *dst = v; }

extern "C" int td_step_record_set(td_step_record *record_dev, uint64_t seed, float lr, float beta1, float beta2, int32_t step,
                                  int32_t ema_mode, float ema_decay, void *stream) {
    TD_REQUIRE(record_dev && step >= 1, "bad argument");
    td_step_record r;
    r.seed = seed;
    r.step_size = (float)((double)lr / (1.0 - pow((double)beta1, (double)step)));
    r.inv_sqrt_bc2 = (float)(1.0 / sqrt(1.0 - pow((double)beta2, (double)step)));
    r.ema_decay = ema_decay;
    r.ema_mode = ema_mode;
    r.reserved = 0;
    hipLaunchKernelGGL(td_step_record_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, record_dev, r);   // the values travel as kernel arguments
    TD_LAUNCH_CHECK();
    return TD_OK;
}

extern "C" int td_clip_adam_ema_rec(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, float *ema, size_t n,
                                    const float *norm, float max_norm, float beta1, float beta2, float eps,
                                    const td_step_record *record, void *stream) {
    TD_REQUIRE(param && grad && exp_avg && exp_avg_sq && ema && record && n > 0, "bad argument");
    TD_REQUIRE(max_norm <= 0.f || norm, "norm pointer required when clipping");
    TdAdamParams a;
    a.p = param; a.g = grad; a.m = exp_avg; a.v = exp_avg_sq; a.ema = ema; a.norm = norm; a.n = n;
    a.max_norm = max_norm; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps;
    a.ema_decay = 0.f; a.ema_mode = 0; a.step_size = 0.f; a.inv_sqrt_bc2 = 1.f;
    a.rec = record;
    size_t b = (n + 255) / 256;
    const int blocks = (int)(b > 4096 ? 4096 : b);
    hipLaunchKernelGGL(td_clip_adam_ema_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
    TD_LAUNCH_CHECK();
    return TD_OK;
}
