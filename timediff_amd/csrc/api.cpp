// Error plumbing and version of libtimediff_hip.so.
#include <stdarg.h>
#include <stdio.h>

#include "../../include/timediff_hip.h"

static thread_local char g_err[512] = "";

void td_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

extern "C" const char *td_last_error(void) { return g_err; }
extern "C" int td_version(void) { return 200; }

// ---------------------------------------------------------------------------------------------
// Explicit kernel-selection knobs (tests and measurements flip them in-process).  No counterpart in the reference.
// ---------------------------------------------------------------------------------------------
static int g_layer_kernel = TD_LAYER_KERNEL_AUTO;
static int g_head_fusion = TD_HEAD_FUSION_AUTO;
int td_head_fusion_mode(void) { return g_head_fusion; }
int td_layer_kernel_mode(void) { return g_layer_kernel; }

extern "C" int td_set_head_fusion(int32_t mode) {
    if (mode != TD_HEAD_FUSION_AUTO && mode != TD_HEAD_FUSION_OFF && mode != TD_HEAD_FUSION_ON) {
        td_set_error("td_set_head_fusion: unknown mode %d", mode);
        return TD_ERR_INVALID;
    }
    g_head_fusion = mode;
    return TD_OK;
}

static int g_side_gemm = 1;
int td_side_gemm_mode(void) { return g_side_gemm; }
extern "C" int td_set_side_gemm(int32_t on) {
    g_side_gemm = on ? 1 : 0;
    return TD_OK;
}

static int g_step_launch = TD_STEP_LAUNCH_AUTO;
int td_step_launch_mode(void) { return g_step_launch; }
extern "C" int td_set_step_launch(int32_t mode) {
    if (mode != TD_STEP_LAUNCH_AUTO && mode != TD_STEP_LAUNCH_OFF && mode != TD_STEP_LAUNCH_ON) {
        td_set_error("td_set_step_launch: unknown mode %d", mode);
        return TD_ERR_INVALID;
    }
    g_step_launch = mode;
    return TD_OK;
}

extern "C" int td_set_layer_kernel(int32_t mode) {
    if (mode != TD_LAYER_KERNEL_AUTO && mode != TD_LAYER_KERNEL_BATCH && mode != TD_LAYER_KERNEL_UNITS && mode != TD_LAYER_KERNEL_CLUSTER) {
        td_set_error("td_set_layer_kernel: unknown mode %d", mode);
        return TD_ERR_INVALID;
    }
    g_layer_kernel = mode;
    return TD_OK;
}

// ---------------------------------------------------------------------------------------------
// Optional per-kernel timing with HIP events on the launch stream (bench.py's roofline probe).
// Not used inside graph capture.  No counterpart in the reference.
// ---------------------------------------------------------------------------------------------
#include <hip/hip_runtime.h>

#define TD_PROF_MAX 1024
static struct {
    int on, count;
    hipEvent_t start[TD_PROF_MAX], stop[TD_PROF_MAX];
    int created;
} g_prof;

extern "C" int td_profile_enable(int on) {
    if (on && !g_prof.created) {
        for (int i = 0; i < TD_PROF_MAX; ++i) {
            if (hipEventCreate(&g_prof.start[i]) != hipSuccess || hipEventCreate(&g_prof.stop[i]) != hipSuccess) {
                td_set_error("hipEventCreate failed");
                return TD_ERR_HIP;
            }
        }
        g_prof.created = 1;
    }
    g_prof.on = on;
    g_prof.count = 0;
    return TD_OK;
}

// called by the recurrent-layer launchers: returns the slot to stop, or -1.  kind: which launches td_profile_enable asked
// for (TD_PROFILE_* bits: inference layers above the first, their saving form, the BPTT sweeps of the layers above the first)
int td_profile_begin(int kind, void *stream) {
    if (!(g_prof.on & kind) || g_prof.count >= TD_PROF_MAX) return -1;
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing((hipStream_t)stream, &st) != hipSuccess || st != hipStreamCaptureStatusNone) return -1;
    int slot = g_prof.count++;
    (void)hipEventRecord(g_prof.start[slot], (hipStream_t)stream);
    return slot;
}
void td_profile_end(int slot, void *stream) {
    if (slot >= 0) (void)hipEventRecord(g_prof.stop[slot], (hipStream_t)stream);
}

extern "C" int td_profile_read(int32_t *count, float *total_ms) {
    float total = 0.f;
    for (int i = 0; i < g_prof.count; ++i) {
        float ms = 0.f;
        if (hipEventSynchronize(g_prof.stop[i]) != hipSuccess || hipEventElapsedTime(&ms, g_prof.start[i], g_prof.stop[i]) != hipSuccess) {
            td_set_error("reading profile event %d failed", i);
            return TD_ERR_HIP;
        }
        total += ms;
    }
    if (count) *count = g_prof.count;
    if (total_ms) *total_ms = total;
    g_prof.count = 0;
    return TD_OK;
}

// ---------------------------------------------------------------------------------------------
// Errors a kernel can only report after the fact.  The cluster-persistent kernels (rnn_cluster.hip, rnn_cluster_bwd.hip)
// hand data between workgroups inside one launch and need all of them resident at once; where that does not hold (a GPU
// shared with another process, CU masking) a bounded poll times out, the kernel poisons its outputs with NaN -- and
// raises this word, which lives in host memory the device can write: td_check_async_errors reads it without touching
// the GPU.  Allocated by the first td_rnn_pack_weights (never under stream capture).  No counterpart in the reference.
// ---------------------------------------------------------------------------------------------
static unsigned *g_async_err = nullptr;
unsigned *td_async_error_word(bool create) {
    if (!g_async_err && create) {
        void *p = nullptr;
        if (hipHostMalloc(&p, 64, hipHostMallocPortable | hipHostMallocMapped) == hipSuccess) {
            g_async_err = (unsigned *)p;
            *(volatile unsigned *)g_async_err = 0;
        } else {
            (void)hipGetLastError();
        }
    }
    return g_async_err;
}

extern "C" int td_check_async_errors(void) {
    if (!g_async_err) return TD_OK;
    const unsigned v = __atomic_exchange_n(g_async_err, 0u, __ATOMIC_RELAXED);
    if (v == 0) return TD_OK;
    td_set_error("a cluster-persistent kernel timed out waiting for another workgroup of its launch (its outputs are NaN): the "
                 "workgroups were not all resident at once -- is the GPU shared with another process or partitioned? "
                 "td_set_layer_kernel(TD_LAYER_KERNEL_UNITS) avoids these kernels");
    return TD_ERR_HIP;
}
