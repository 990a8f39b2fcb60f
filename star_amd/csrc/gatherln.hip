// Gather-at-output of the semantic encoder (second half).
//
// Reference: the semantic encoder (a GPT-2 stack) ends with  hidden = self.ln_f(hidden)  over ALL T frames
// (src/models/my_transformer.py:945) and the caller keeps only the nugget rows of the result,
//     semantic_repr = semantic_repr[arange(B)[:, None], nugget_indices]          (w2v_asr.py:566-567, w2v_simul.py:446-447)
// LayerNorm is row-wise, so ln_f(hidden)[idx] == ln_f(hidden[idx]): here the rows are gathered FIRST and only the
// B*K gathered rows are normalised -- the [B,T,d] ln_f output is never written or read back (at configs[1]: 98 MB
// written + 98 MB read in the forward, and a LayerNorm backward over 48000 rows instead of 12000).
//   forward : out[b,j,:] = (x - mean) * rstd * gamma + beta,  x = src[b, frame(idx[b,j]), :]   (fp32 statistics, two-pass)
//             mean / rstd [B,K] are kept for the backward
//   backward: per gathered row  g = dout * gamma,  xhat = (x - mean) * rstd,
//             drow = rstd * (g - mean_d(g) - xhat * mean_d(g * xhat))      -> drows [B,K,d]
//             which star_gather_rows_bwd then adds into d src (every d src row written once, duplicates summed);
//             dgamma = sum_rows dout * xhat, dbeta = sum_rows dout: row-block partials + a fixed-order finish.
// One wave per row, 16 B per lane per step, rows of up to 4096 elements stay in registers.  HBM-bound:
//   forward reads and writes row_bytes * B*K; backward reads 2 * row_bytes * B*K, writes row_bytes * B*K.
#include "common.h"

namespace star {

__device__ __forceinline__ int64_t gln_frame(int64_t t, int T) {      // same rule as gather.hip: wrap, then clamp
    if (t < 0) t += T;
    if (t < 0) t = 0;
    if (t >= T) t = T - 1;
    return t;
}

template <typename T_ELEM>
__device__ __forceinline__ void gln_load8(const T_ELEM* p, float (&v)[8]);
template <>
__device__ __forceinline__ void gln_load8<__bf16>(const __bf16* p, float (&v)[8]) {
    const bf16x8 r = *(const bf16x8*)p;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (float)r[i];
}
template <>
__device__ __forceinline__ void gln_load8<float>(const float* p, float (&v)[8]) {
    const f32x4 a = *(const f32x4*)p, b = *(const f32x4*)(p + 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        v[i] = a[i];
        v[4 + i] = b[i];
    }
}
template <typename T_ELEM>
__device__ __forceinline__ void gln_store8(T_ELEM* p, const float (&v)[8]);
template <>
__device__ __forceinline__ void gln_store8<__bf16>(__bf16* p, const float (&v)[8]) {
    bf16x8 r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r[i] = (__bf16)v[i];
    *(bf16x8*)p = r;
}
template <>
__device__ __forceinline__ void gln_store8<float>(float* p, const float (&v)[8]) {
    *(f32x4*)p = (f32x4){v[0], v[1], v[2], v[3]};
    *(f32x4*)(p + 4) = (f32x4){v[4], v[5], v[6], v[7]};
}

// CH = chunks of 8 elements per lane (d <= 512 * CH)
template <typename T_ELEM, int CH>
__global__ __launch_bounds__(256) void gather_ln_fwd_kernel(const T_ELEM* __restrict__ src, const int64_t* __restrict__ idx,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            float eps, int T, int K, int d, int64_t rows_total,
                                                            T_ELEM* __restrict__ out, float* __restrict__ mean_out,
                                                            float* __restrict__ rstd_out) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows_total) return;
    const int64_t b = row / K;
    const T_ELEM* s = src + ((size_t)b * T + gln_frame(idx[row], T)) * d;
    float x[CH][8];
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int e = (c * 64 + lane) * 8;
        if (e < d) {
            gln_load8<T_ELEM>(s + e, x[c]);
#pragma unroll
            for (int i = 0; i < 8; ++i) sum += x[c][i];
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) x[c][i] = 0.f;
        }
    }
    const float mean = wave_sum(sum) / (float)d;
    float sq = 0.f;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int e = (c * 64 + lane) * 8;
        if (e < d) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float t = x[c][i] - mean;
                sq += t * t;
            }
        }
    }
    const float rstd = rsqrtf(wave_sum(sq) / (float)d + eps);
    T_ELEM* o = out + (size_t)row * d;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int e = (c * 64 + lane) * 8;
        if (e < d) {
            float g[8], bt[8], y[8];
            gln_load8<float>(gamma + e, g);
            gln_load8<float>(beta + e, bt);
#pragma unroll
            for (int i = 0; i < 8; ++i) y[i] = (x[c][i] - mean) * rstd * g[i] + bt[i];
            gln_store8<T_ELEM>(o + e, y);
        }
    }
    if (lane == 0) {
        mean_out[row] = mean;
        rstd_out[row] = rstd;
    }
}

constexpr int GLN_ROWS_PER_WAVE = 8;                          // rows one wave walks in the backward
constexpr int GLN_ROWS_PER_WG = 4 * GLN_ROWS_PER_WAVE;

template <typename T_ELEM, int CH>
__global__ __launch_bounds__(256) void gather_ln_bwd_kernel(const T_ELEM* __restrict__ src, const int64_t* __restrict__ idx,
                                                            const float* __restrict__ gamma, const float* __restrict__ mean_in,
                                                            const float* __restrict__ rstd_in, const T_ELEM* __restrict__ dout,
                                                            int T, int K, int d, int64_t rows_total,
                                                            T_ELEM* __restrict__ drows, float* __restrict__ partial) {
    extern __shared__ float gln_red[];                         // [4 waves][2][d]
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    float gm[CH][8], dg[CH][8], db[CH][8];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int e = (c * 64 + lane) * 8;
        if (e < d) gln_load8<float>(gamma + e, gm[c]);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            dg[c][i] = 0.f;
            db[c][i] = 0.f;
            if (e >= d) gm[c][i] = 0.f;
        }
    }
    const int64_t row0 = (int64_t)blockIdx.x * GLN_ROWS_PER_WG + wid * GLN_ROWS_PER_WAVE;
    for (int r = 0; r < GLN_ROWS_PER_WAVE; ++r) {
        const int64_t row = row0 + r;
        if (row >= rows_total) break;
        const int64_t b = row / K;
        const T_ELEM* s = src + ((size_t)b * T + gln_frame(idx[row], T)) * d;
        const T_ELEM* go = dout + (size_t)row * d;
        const float mean = mean_in[row], rstd = rstd_in[row];
        float xh[CH][8], g[CH][8];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int e = (c * 64 + lane) * 8;
            if (e < d) {
                float x[8], dy[8];
                gln_load8<T_ELEM>(s + e, x);
                gln_load8<T_ELEM>(go + e, dy);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    xh[c][i] = (x[i] - mean) * rstd;
                    g[c][i] = dy[i] * gm[c][i];
                    s1 += g[c][i];
                    s2 += g[c][i] * xh[c][i];
                    dg[c][i] += dy[i] * xh[c][i];
                    db[c][i] += dy[i];
                }
            }
        }
        const float c1 = wave_sum(s1) / (float)d, c2 = wave_sum(s2) / (float)d;
        T_ELEM* o = drows + (size_t)row * d;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int e = (c * 64 + lane) * 8;
            if (e < d) {
                float dx[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) dx[i] = rstd * (g[c][i] - c1 - xh[c][i] * c2);
                gln_store8<T_ELEM>(o + e, dx);
            }
        }
    }
    // column partials of the workgroup's rows: the four waves' sums are added in wave order
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int e = (c * 64 + lane) * 8;
        if (e < d) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                gln_red[(wid * 2 + 0) * d + e + i] = dg[c][i];
                gln_red[(wid * 2 + 1) * d + e + i] = db[c][i];
            }
        }
    }
    __syncthreads();
    for (int n = threadIdx.x; n < 2 * d; n += 256) {
        const int which = n / d, col = n - which * d;
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) t += gln_red[(w * 2 + which) * d + col];
        partial[((size_t)blockIdx.x * 2 + which) * d + col] = t;
    }
}

// dgamma / dbeta = sum over row blocks in a fixed order: 64 columns per workgroup, four wave-sized groups take the
// blocks k = g, g + 4, ... (coalesced 256-byte rows, four loads in flight per thread), then the groups are added in order
__global__ __launch_bounds__(256) void gather_ln_finish_kernel(const float* __restrict__ partial, int nblk, int d,
                                                               float* __restrict__ dgamma, float* __restrict__ dbeta) {
    __shared__ float red[4][64];
    const int n = blockIdx.x * 64 + (threadIdx.x & 63), g = threadIdx.x >> 6;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    if (n < 2 * d) {
        const int which = n / d, col = n - which * d;
        const float* p = partial + (size_t)which * d + col;
        int k = g;
        for (; k + 12 < nblk; k += 16) {
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[u] += p[(size_t)(k + 4 * u) * 2 * d];
        }
        for (; k < nblk; k += 4) acc[0] += p[(size_t)k * 2 * d];
    }
    red[g][threadIdx.x & 63] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    __syncthreads();
    if (g == 0 && n < 2 * d) {
        const int which = n / d, col = n - which * d;
        const int l = threadIdx.x;
        (which ? dbeta : dgamma)[col] = (red[0][l] + red[1][l]) + (red[2][l] + red[3][l]);
    }
}

template <typename T_ELEM>
static int launch_gln_fwd(const void* src, const int64_t* idx, const float* gamma, const float* beta, float eps, int B,
                          int T, int K, int d, void* out, float* mean, float* rstd, hipStream_t st) {
    const int64_t rows = (int64_t)B * K;
    const dim3 grid(cdiv(rows, 4)), blk(256);
#define GLN_FWD(CH)                                                                                          \
    hipLaunchKernelGGL((gather_ln_fwd_kernel<T_ELEM, CH>), grid, blk, 0, st, (const T_ELEM*)src, idx, gamma, beta, eps, T, \
                       K, d, rows, (T_ELEM*)out, mean, rstd)
    if (d <= 512) GLN_FWD(1);
    else if (d <= 1024) GLN_FWD(2);
    else if (d <= 2048) GLN_FWD(4);
    else GLN_FWD(8);
#undef GLN_FWD
    return STAR_OK;
}

template <typename T_ELEM>
static int launch_gln_bwd(const void* src, const int64_t* idx, const float* gamma, const float* mean, const float* rstd,
                          const void* dout, int B, int T, int K, int d, void* drows, float* partial, hipStream_t st) {
    const int64_t rows = (int64_t)B * K;
    const dim3 grid(cdiv(rows, GLN_ROWS_PER_WG)), blk(256);
    const size_t smem = (size_t)8 * d * sizeof(float);
#define GLN_BWD(CH)                                                                                          \
    hipLaunchKernelGGL((gather_ln_bwd_kernel<T_ELEM, CH>), grid, blk, smem, st, (const T_ELEM*)src, idx, gamma, mean, rstd, \
                       (const T_ELEM*)dout, T, K, d, rows, (T_ELEM*)drows, partial)
    if (d <= 512) GLN_BWD(1);
    else if (d <= 1024) GLN_BWD(2);
    else GLN_BWD(4);
#undef GLN_BWD
    return STAR_OK;
}

}  // namespace star

extern "C" size_t star_gather_layernorm_bwd_workspace(int B, int K, int d) {
    return (size_t)star::cdiv((int64_t)B * K, star::GLN_ROWS_PER_WG) * 2 * d * sizeof(float);
}

extern "C" int star_gather_layernorm_fwd(const void* src, int dtype, const int64_t* idx, const float* gamma,
                                         const float* beta, float eps, int B, int T, int K, int d, void* out, float* mean,
                                         float* rstd, void* stream) {
    using namespace star;
    STAR_REQUIRE(src && idx && gamma && beta && out && mean && rstd, "star_gather_layernorm_fwd: null pointer");
    STAR_REQUIRE(B >= 1 && T >= 1 && K >= 1, "star_gather_layernorm_fwd: bad shape B=%d T=%d K=%d", B, T, K);
    STAR_REQUIRE(d >= 8 && d % 8 == 0 && d <= 4096, "star_gather_layernorm_fwd: d=%d must be a multiple of 8, at most 4096", d);
    STAR_REQUIRE(dtype == STAR_DT_BF16 || dtype == STAR_DT_F32, "star_gather_layernorm_fwd: dtype %d", dtype);
    STAR_REQUIRE(eps > 0.f, "star_gather_layernorm_fwd: eps must be positive");
    if (dtype == STAR_DT_BF16)
        launch_gln_fwd<__bf16>(src, idx, gamma, beta, eps, B, T, K, d, out, mean, rstd, (hipStream_t)stream);
    else
        launch_gln_fwd<float>(src, idx, gamma, beta, eps, B, T, K, d, out, mean, rstd, (hipStream_t)stream);
    STAR_LAUNCH_CHECK("star_gather_layernorm_fwd");
    return STAR_OK;
}

extern "C" int star_gather_layernorm_bwd(const void* src, int dtype, const int64_t* idx, const float* gamma,
                                         const float* mean, const float* rstd, const void* dout, int B, int T, int K,
                                         int d, void* drows, float* dgamma, float* dbeta, void* workspace,
                                         size_t workspace_bytes, void* stream) {
    using namespace star;
    STAR_REQUIRE(src && idx && gamma && mean && rstd && dout && drows && dgamma && dbeta && workspace,
                 "star_gather_layernorm_bwd: null pointer");
    STAR_REQUIRE(B >= 1 && T >= 1 && K >= 1, "star_gather_layernorm_bwd: bad shape B=%d T=%d K=%d", B, T, K);
    STAR_REQUIRE(d >= 8 && d % 8 == 0 && d <= 1536, "star_gather_layernorm_bwd: d=%d must be a multiple of 8, at most 1536", d);
    STAR_REQUIRE(dtype == STAR_DT_BF16 || dtype == STAR_DT_F32, "star_gather_layernorm_bwd: dtype %d", dtype);
    if (workspace_bytes < star_gather_layernorm_bwd_workspace(B, K, d)) {
        set_error("star_gather_layernorm_bwd: workspace %zu < %zu bytes", workspace_bytes,
                  star_gather_layernorm_bwd_workspace(B, K, d));
        return STAR_ERR_WORKSPACE;
    }
    const int nblk = cdiv((int64_t)B * K, GLN_ROWS_PER_WG);
    if (dtype == STAR_DT_BF16)
        launch_gln_bwd<__bf16>(src, idx, gamma, mean, rstd, dout, B, T, K, d, drows, (float*)workspace, (hipStream_t)stream);
    else
        launch_gln_bwd<float>(src, idx, gamma, mean, rstd, dout, B, T, K, d, drows, (float*)workspace, (hipStream_t)stream);
    hipLaunchKernelGGL(gather_ln_finish_kernel, dim3(cdiv(2 * d, 64)), dim3(256), 0, (hipStream_t)stream,
                       (const float*)workspace, nblk, d, dgamma, dbeta);
    STAR_LAUNCH_CHECK("star_gather_layernorm_bwd");
    return STAR_OK;
}
