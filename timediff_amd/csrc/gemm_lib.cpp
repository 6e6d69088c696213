// Weight-gradient GEMMs through rocBLAS: d W (M x N, fp32, accumulated) += A (M x K, bf16) . B (N x K, bf16)^T is a plain
// library GEMM (K = L * batch columns, up to hundreds of thousands), the one place on this path where a vendor GEMM
// is the right tool; td_dw_gemm_kernel (rnn_train.hip) stays for the shapes that need its column remap (hidden
// sizes that are not a multiple of 32, the first layer's few input channels).
#include <rocblas/rocblas.h>

#include "common.h"

static rocblas_handle g_blas = nullptr;

// C[m][n] += sum_k A[m][k] B[n][k]; A, B row-major bf16 with k contiguous (lda, ldb), C row-major fp32 (ldc).
// One path per shape: the caller (td_dw_gemm, rnn_train.hip) decides from the shape alone whether this runs; a failure
// here is an error, never a silent switch to another kernel.
// nsplit > 1: the contraction index is cut into nsplit equal chunks that run as the batches of ONE strided-batched GEMM,
// each writing its own M x N partial result (row-major, packed) into `partial`; the caller adds them to C.  With K in
// the hundreds of thousands and M x N of only a few 256 x 256 tiles this is what fills the chip.
static int td_blas_ready(hipStream_t stream) {
    if (!g_blas) {
        if (rocblas_create_handle(&g_blas) != rocblas_status_success) {
            g_blas = nullptr;
            td_set_error("rocblas_create_handle failed");
            return TD_ERR_HIP;
        }
    }
    if (rocblas_set_stream(g_blas, stream) != rocblas_status_success) {
        td_set_error("rocblas_set_stream failed");
        return TD_ERR_HIP;
    }
    return TD_OK;
}

// dw[r][k] (R x K row-major fp32, accumulated) += sum_i left[i][r] * right[i][k], left / right row-major fp32 with row strides
// lstride / rstride: the weight gradients of the time-embedding MLP and the FiLM linear over a large batch of rows
// (rnn.hip: td_rnn_time_film_backward).  Column-major view: dw' (K x R, ld K) = right' (K x n) . left'^T (n x R).
int td_outer_sum_lib(const float *left, size_t lstride, const float *right, size_t rstride, float *dw, int R, int K, int n,
                     hipStream_t stream) {
    int rc = td_blas_ready(stream);
    if (rc != TD_OK) return rc;
    const float one = 1.0f;
    rocblas_status st = rocblas_sgemm(g_blas, rocblas_operation_none, rocblas_operation_transpose, K, R, n, &one, right,
                                      (rocblas_int)rstride, left, (rocblas_int)lstride, &one, dw, K);
    if (st != rocblas_status_success) {
        td_set_error("rocblas_sgemm failed with status %d (R=%d K=%d n=%d)", (int)st, R, K, n);
        return TD_ERR_HIP;
    }
    return TD_OK;
}

int td_dw_gemm_lib(const void *A, size_t lda, const void *B, size_t ldb, float *C, int ldc, int M, int N, int K,
                   int nsplit, float *partial, hipStream_t stream) {
    int rc = td_blas_ready(stream);
    if (rc != TD_OK) return rc;
    const float one = 1.0f, zero = 0.0f;
    if (nsplit > 1) {
        const int kc = K / nsplit;
        rocblas_status st = rocblas_gemm_strided_batched_ex(
            g_blas, rocblas_operation_transpose, rocblas_operation_none, N, M, kc, &one, B, rocblas_datatype_bf16_r, (rocblas_int)ldb,
            (rocblas_stride)kc, A, rocblas_datatype_bf16_r, (rocblas_int)lda, (rocblas_stride)kc, &zero, partial, rocblas_datatype_f32_r, N,
            (rocblas_stride)M * N, partial, rocblas_datatype_f32_r, N, (rocblas_stride)M * N, nsplit, rocblas_datatype_f32_r,
            rocblas_gemm_algo_standard, 0, 0);
        if (st != rocblas_status_success) {
            td_set_error("rocblas_gemm_strided_batched_ex failed with status %d (M=%d N=%d K=%d x %d)", (int)st, M, N, kc, nsplit);
            return TD_ERR_HIP;
        }
        return TD_OK;
    }
    // column-major view: C' (N x M, ld ldc) = Bc^T (N x K) . Ac (K x M), Bc = B as (K x N, ld ldb), Ac = A as (K x M, ld lda)
    rocblas_status st = rocblas_gemm_ex(g_blas, rocblas_operation_transpose, rocblas_operation_none, N, M, K, &one, B,
                                        rocblas_datatype_bf16_r, (rocblas_int)ldb, A, rocblas_datatype_bf16_r,
                                        (rocblas_int)lda, &one, C, rocblas_datatype_f32_r, ldc, C, rocblas_datatype_f32_r, ldc,
                                        rocblas_datatype_f32_r, rocblas_gemm_algo_standard, 0, 0);
    if (st != rocblas_status_success) {
        td_set_error("rocblas_gemm_ex failed with status %d (M=%d N=%d K=%d)", (int)st, M, N, K);
        return TD_ERR_HIP;
    }
    return TD_OK;
}
