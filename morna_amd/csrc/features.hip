// features.hip -- feature-hashed TF-IDF matrix build on gfx950.
//
// Replaces the add_junction loop and the add_item hand-off of the reference
// (commanderson/morna morna.py:344-388, 405-424).  Parity contract: the fp32
// matrix is BIT-IDENTICAL to the reference's, which sums fp64 contributions per
// (sample, column) cell in file order and rounds once to fp32.
//
// Data flow (all in HBM, one pass over the nnz stream):
//   hash_keys_kernel     J keys -> column, sign*idf
//   line_flags_kernel    marks lines whose id list is not strictly monotone
//                        (the only lines that could repeat a sample)
//   accumulate_kernel    one workgroup per feature COLUMN: walks the junction
//                        lines that hash to its column in file order and adds
//                        cov*idf into an fp64 column image [N] that stays in L2;
//                        distinct columns never share a cell, lines of one
//                        column are serialised by the workgroup barrier, so the
//                        per-cell order is the file order with no atomics.
//   transpose_convert    fp64 [D][N] -> fp32 [N][dpad] through an LDS tile
//   row_norms_kernel     canonical dot(x, x) per row
#include <algorithm>
#include <cstring>

#include "common.hpp"
#include "devutil.hpp"

namespace morna {

// ------------------------------------------------------------------ hash pass

__global__ void hash_keys_kernel(const uint8_t *__restrict__ keys, const int64_t *__restrict__ key_off,
                                 int64_t J, int32_t dim, const double *__restrict__ idf,
                                 int32_t *__restrict__ col_out, double *__restrict__ sidf_out,
                                 int32_t *__restrict__ hash_out, int32_t *__restrict__ sign_out)
{
    int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= J) return;
    int64_t b = key_off[j], e = key_off[j + 1];
    int32_t h = (int32_t)murmur3_32(keys + b, e - b, 0u);   // mmh3.hash, morna.py:369
    int32_t col = floored_mod(h, dim);                      // morna.py:371
    col_out[j] = col;
    if (sidf_out) sidf_out[j] = h < 0 ? -idf[j] : idf[j];   // multiplier * (cov * idf): sign is exact
    if (hash_out) hash_out[j] = h;
    if (sign_out) sign_out[j] = h < 0 ? -1 : 1;             // morna.py:370, before the modulo
}

// One wave per line: flag lines whose item ids are not strictly monotone.
__global__ void line_flags_kernel(const int64_t *__restrict__ row_ptr, const int32_t *__restrict__ ids,
                                  int64_t J, uint8_t *__restrict__ serial_out)
{
    const int lane = threadIdx.x & (WAVE - 1);
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / WAVE;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) / WAVE;
    for (int64_t j = wave; j < J; j += nwaves) {
        int64_t b = row_ptr[j], e = row_ptr[j + 1];
        int inc = 1, dec = 1;
        for (int64_t t = b + lane; t + 1 < e; t += WAVE) {
            int32_t a = ids[t], c = ids[t + 1];
            inc &= (a < c);
            dec &= (a > c);
        }
        inc = __all(inc);
        dec = __all(dec);
        if (lane == 0) serial_out[j] = (inc || dec) ? 0 : 1;
    }
}

// ------------------------------------------------------------ accumulate pass

#define ACC_THREADS 256

__global__ __launch_bounds__(ACC_THREADS) void accumulate_kernel(
    const int32_t *__restrict__ col, const double *__restrict__ sidf, const uint8_t *__restrict__ serial,
    const int64_t *__restrict__ row_ptr, const int32_t *__restrict__ ids, const int32_t *__restrict__ cov,
    int64_t J, int64_t n_items, double *__restrict__ colacc /* [D][n_items] */)
{
    __shared__ int s_list[ACC_THREADS];
    __shared__ int s_wcnt[ACC_THREADS / WAVE];
    const int c = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), w = tid / WAVE;
    double *acc = colacc + (int64_t)c * n_items;

    for (int64_t j0 = 0; j0 < J; j0 += ACC_THREADS) {
        const int64_t j = j0 + tid;
        const bool m = j < J && col[j] == c;
        // ordered compaction of this chunk's matching lines
        const unsigned long long bal = __ballot(m);
        if (lane == 0) s_wcnt[w] = __popcll(bal);
        __syncthreads();
        int before = 0, total = 0;
#pragma unroll
        for (int i = 0; i < ACC_THREADS / WAVE; i++) {
            int n = s_wcnt[i];
            if (i < w) before += n;
            total += n;
        }
        if (m) s_list[before + __popcll(bal & ((1ull << lane) - 1ull))] = tid;
        __syncthreads();
        for (int i = 0; i < total; i++) {
            const int64_t jj = j0 + s_list[i];
            const int64_t b = row_ptr[jj], e = row_ptr[jj + 1];
            const double wgt = sidf[jj];
            if (serial[jj]) {
                // a sample may repeat inside this line: keep the line's own order
                if (tid == 0)
                    for (int64_t t = b; t < e; t++) {
                        int32_t id = ids[t];
                        acc[id] = __dadd_rn(acc[id], __dmul_rn((double)cov[t], wgt));
                    }
            } else {
                for (int64_t t = b + tid; t < e; t += ACC_THREADS) {
                    int32_t id = ids[t];
                    // tf_idf = cov * idf (one rounding), then += (one rounding): morna.py:384-388
                    acc[id] = __dadd_rn(acc[id], __dmul_rn((double)cov[t], wgt));
                }
            }
            __syncthreads();   // the next line of this column may touch the same cells
        }
        __syncthreads();       // s_list / s_wcnt reuse
    }
}

// --------------------------------------------------- fp64 [D][N] -> fp32 [N][dpad]

#define TT 64
__global__ __launch_bounds__(256) void transpose_convert_kernel(const double *__restrict__ colacc,
                                                                int64_t n_items, int32_t dim, int32_t dpad,
                                                                float *__restrict__ X)
{
    __shared__ float tile[TT][TT + 1];
    const int64_t n0 = (int64_t)blockIdx.x * TT;
    const int32_t c0 = blockIdx.y * TT;
    const int tx = threadIdx.x & (TT - 1), ty = threadIdx.x / TT;   // 64 x 4
    for (int r = ty; r < TT; r += 4) {
        int32_t c = c0 + r;
        int64_t n = n0 + tx;
        float v = 0.f;
        if (c < dim && n < n_items) v = __double2float_rn(colacc[(int64_t)c * n_items + n]);  // add_item cast
        tile[r][tx] = v;
    }
    __syncthreads();
    for (int r = ty; r < TT; r += 4) {
        int64_t n = n0 + r;
        int32_t c = c0 + tx;
        if (n < n_items && c < dpad) X[n * dpad + c] = tile[tx][r];
    }
}

// ------------------------------------------------------------------ row norms

__global__ __launch_bounds__(256) void row_norms_kernel(const float *__restrict__ X, int64_t n_items,
                                                        int32_t dpad, float *__restrict__ norm2)
{
    const int lane = threadIdx.x & (WAVE - 1);
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / WAVE;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) / WAVE;
    for (int64_t r = wave; r < n_items; r += nwaves) {
        const float4 *x = (const float4 *)(X + r * dpad);
        float d = wave_dot(x, x, dpad / 4, lane);
        if (lane == 0) norm2[r] = d;
    }
}

// ------------------------------------------------------------------ host side

int compute_norms(morna_index *h)
{
    MORNA_TRY(h->norm2.alloc((size_t)h->n_items));
    if (h->n_items > 0) {
        int64_t waves = h->n_items;
        int blocks = (int)std::min<int64_t>((waves + 3) / 4, 256 * 16);
        hipLaunchKernelGGL(row_norms_kernel, dim3(blocks), dim3(256), 0, h->stream, h->X.p, h->n_items,
                           h->dpad, h->norm2.p);
        HIP_TRY(hipGetLastError());
    }
    h->norms_valid = true;
    return MORNA_OK;
}

int upload_host_rows(morna_index *h)
{
    if (!h->host_dirty) return MORNA_OK;
    const int64_t n = h->host_n;
    MORNA_TRY(h->X.alloc((size_t)std::max<int64_t>(n, 1) * h->dpad));
    HIP_TRY(hipMemsetAsync(h->X.p, 0, (size_t)std::max<int64_t>(n, 1) * h->dpad * sizeof(float), h->stream));
    if (n > 0)
        HIP_TRY(hipMemcpy2DAsync(h->X.p, (size_t)h->dpad * sizeof(float), h->host_rows.data(),
                                 (size_t)h->dim * sizeof(float), (size_t)h->dim * sizeof(float), (size_t)n,
                                 hipMemcpyHostToDevice, h->stream));
    h->n_items = n;
    h->host_dirty = false;
    h->built = false;
    MORNA_TRY(compute_norms(h));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return MORNA_OK;
}

int hash_keys_device(morna_index *h, const uint8_t *key_bytes, const int64_t *key_off, int64_t J,
                     int32_t *hash_out, int32_t *col_out, int32_t *sign_out)
{
    if (J <= 0) return MORNA_OK;
    const int64_t nbytes = key_off[J];
    DevBuf<uint8_t> dk;
    DevBuf<int64_t> doff;
    DevBuf<int32_t> dh, dc, ds;
    int rc = MORNA_OK;
    if ((rc = dk.alloc((size_t)nbytes)) || (rc = doff.alloc((size_t)J + 1)) || (rc = dh.alloc((size_t)J)) ||
        (rc = dc.alloc((size_t)J)) || (rc = ds.alloc((size_t)J)))
        goto done;
#define HASH_TRY(e)                                                  \
    if ((e) != hipSuccess) {                                         \
        set_error("%s failed: %s", #e, hipGetErrorString(hipGetLastError())); \
        rc = MORNA_E_HIP;                                            \
        goto done;                                                   \
    }
    HASH_TRY(hipMemcpyAsync(dk.p, key_bytes, (size_t)nbytes, hipMemcpyHostToDevice, h->stream));
    HASH_TRY(hipMemcpyAsync(doff.p, key_off, (size_t)(J + 1) * sizeof(int64_t), hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(hash_keys_kernel, dim3((unsigned)((J + 255) / 256)), dim3(256), 0, h->stream, dk.p, doff.p,
                       J, h->dim, (const double *)nullptr, dc.p, (double *)nullptr, dh.p, ds.p);
    HASH_TRY(hipGetLastError());
    if (hash_out) HASH_TRY(hipMemcpyAsync(hash_out, dh.p, (size_t)J * 4, hipMemcpyDeviceToHost, h->stream));
    if (col_out) HASH_TRY(hipMemcpyAsync(col_out, dc.p, (size_t)J * 4, hipMemcpyDeviceToHost, h->stream));
    if (sign_out) HASH_TRY(hipMemcpyAsync(sign_out, ds.p, (size_t)J * 4, hipMemcpyDeviceToHost, h->stream));
    HASH_TRY(hipStreamSynchronize(h->stream));
#undef HASH_TRY
done:
    dk.release();
    doff.release();
    dh.release();
    dc.release();
    ds.release();
    return rc;
}

int build_features(morna_index *h, int64_t n_items)
{
    if (!h->staged) {
        set_error("build_features: no junction lines staged");
        return MORNA_E_STATE;
    }
    if (n_items <= 0) {
        set_error("no internal ids were assigned: no sample passes the sample threshold");
        return MORNA_E_EMPTY;
    }
    const int64_t J = h->J;
    const int32_t D = h->dim;
    DevBuf<int32_t> col;
    DevBuf<double> sidf, colacc;
    DevBuf<uint8_t> serial;
    int rc = MORNA_OK;
    // algorithmic bytes of this pass: 8*nnz + keys + 8*J + 4*N*D
    const int64_t alg_bytes = 8 * h->nnz + h->key_bytes_n + 8 * J + 4 * n_items * (int64_t)D;
    if ((rc = col.alloc((size_t)J)) || (rc = sidf.alloc((size_t)J)) || (rc = serial.alloc((size_t)J)) ||
        (rc = colacc.alloc((size_t)D * (size_t)n_items)) || (rc = h->X.alloc((size_t)n_items * h->dpad)))
        goto done;
    {
        ScopedTimer tm(h, MORNA_T_FEATURES, alg_bytes);
        hipError_t e;
        if ((e = hipMemsetAsync(colacc.p, 0, (size_t)D * (size_t)n_items * sizeof(double), h->stream)) != hipSuccess) {
            set_error("memset colacc: %s", hipGetErrorString(e));
            rc = MORNA_E_HIP;
            goto done;
        }
        if (J > 0) {
            hipLaunchKernelGGL(hash_keys_kernel, dim3((unsigned)((J + 255) / 256)), dim3(256), 0, h->stream,
                               h->s_keys.p, h->s_key_off.p, J, D, h->s_idf.p, col.p, sidf.p, (int32_t *)nullptr,
                               (int32_t *)nullptr);
            int fl_blocks = (int)std::min<int64_t>((J + 3) / 4, 256 * 8);
            hipLaunchKernelGGL(line_flags_kernel, dim3(fl_blocks), dim3(256), 0, h->stream, h->s_row_ptr.p,
                               h->s_ids.p, J, serial.p);
            hipLaunchKernelGGL(accumulate_kernel, dim3(D), dim3(ACC_THREADS), 0, h->stream, col.p, sidf.p, serial.p,
                               h->s_row_ptr.p, h->s_ids.p, h->s_cov.p, J, n_items, colacc.p);
        }
        dim3 tg((unsigned)((n_items + TT - 1) / TT), (unsigned)((h->dpad + TT - 1) / TT));
        hipLaunchKernelGGL(transpose_convert_kernel, tg, dim3(256), 0, h->stream, colacc.p, n_items, D, h->dpad,
                           h->X.p);
        if ((e = hipGetLastError()) != hipSuccess) {
            set_error("feature kernels: %s", hipGetErrorString(e));
            rc = MORNA_E_HIP;
            goto done;
        }
        h->n_items = n_items;
        h->host_n = 0;
        h->host_rows.clear();
        h->host_dirty = false;
        h->built = false;
        if ((rc = compute_norms(h))) goto done;
    }
    {
        hipError_t e = hipStreamSynchronize(h->stream);   // scratch buffers are freed below
        if (e != hipSuccess) {
            set_error("build_features: %s", hipGetErrorString(e));
            rc = MORNA_E_HIP;
        }
    }
done:
    col.release();
    sidf.release();
    serial.release();
    colacc.release();
    return rc;
}

}  // namespace morna
