// Nugget type embedding added to the encoder states (first half):
//     inputs_embed = encoder_out.last_hidden_state + semantic_encoder.type_embed(type_ind_table)
// (src/models/w2v_asr.py:383,400,456,560; w2v_simul.py:432; type_embed = nn.Embedding(2, d), my_transformer.py:761).
// The reference materialises the [B,T,d] embedding lookup and then adds it (3 x [B,T,d] of traffic more than needed);
// here one pass reads X and writes X + E[table]: algorithmic bytes 2 * N * d * sizeof(T) (+ 8 N for the table).
// Backward: dX = dout (identity, no kernel), dE[c] = sum over rows with table == c of dout: a column reduction in
// two deterministic stages (row-block partials in a fixed order, then a fixed-order sum over the blocks).
#include "common.h"

namespace star {

template <typename T> struct TeVec;
template <> struct TeVec<__bf16> {
    static constexpr int N = 8;
    typedef bf16x8 type;
};
template <> struct TeVec<float> {
    static constexpr int N = 4;
    typedef f32x4 type;
};

template <typename T>
__global__ __launch_bounds__(256) void type_embed_add_kernel(const T* __restrict__ x, const int64_t* __restrict__ table,
                                                             const T* __restrict__ emb, int64_t nvec, int vec_per_row,
                                                             T* __restrict__ out) {
    using V = typename TeVec<T>::type;
    constexpr int N = TeVec<T>::N;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= nvec) return;
    const int64_t row = i / vec_per_row;
    const int c = (int)(i - row * vec_per_row);
    const int ty = table[row] != 0 ? 1 : 0;
    const V xv = ((const V*)x)[i];
    const V ev = ((const V*)emb)[(int64_t)ty * vec_per_row + c];
    V o;
#pragma unroll
    for (int j = 0; j < N; ++j) o[j] = (T)((float)xv[j] + (float)ev[j]);       // one rounding, as torch's add
    ((V*)out)[i] = o;
}

// stage 1: workgroup (column tile of 64 vectors, row block) -> part[rb][2][d]
template <typename T>
__global__ __launch_bounds__(256) void type_embed_grad_kernel(const T* __restrict__ dout,
                                                              const int64_t* __restrict__ table, int64_t N_rows, int d,
                                                              int rows_per_block, float* __restrict__ part) {
    using V = typename TeVec<T>::type;
    constexpr int N = TeVec<T>::N;
    __shared__ float red[4][2][64 * N];
    const int tx = threadIdx.x & 63;
    const int ty = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int col = (blockIdx.x * 64 + tx) * N;
    const bool col_ok = col < d;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
    const int64_t r1 = min(r0 + rows_per_block, N_rows);
    float a0[N], a1[N];
#pragma unroll
    for (int j = 0; j < N; ++j) { a0[j] = 0.f; a1[j] = 0.f; }
    if (col_ok) {
        for (int64_t r = r0 + ty; r < r1; r += 4) {
            const bool one = table[r] != 0;                                   // wave-uniform (scalar load)
            const V g = *(const V*)(dout + r * d + col);
#pragma unroll
            for (int j = 0; j < N; ++j) {
                const float v = (float)g[j];
                a0[j] += one ? 0.f : v;
                a1[j] += one ? v : 0.f;
            }
        }
    }
#pragma unroll
    for (int j = 0; j < N; ++j) {
        red[ty][0][tx * N + j] = a0[j];
        red[ty][1][tx * N + j] = a1[j];
    }
    __syncthreads();
    // 256 threads sum the 4 row phases in a fixed order: 2 * 64 * N outputs
    for (int o = threadIdx.x; o < 2 * 64 * N; o += 256) {
        const int c = o / (64 * N), e = o % (64 * N);
        const int cc = blockIdx.x * 64 * N + e;
        if (cc < d) part[((int64_t)blockIdx.y * 2 + c) * d + cc] = ((red[0][c][e] + red[1][c][e]) + red[2][c][e]) + red[3][c][e];
    }
}

// stage 2: demb[c][col] = sum over the row blocks, fixed order: 64 outputs per workgroup, the blocks dealt round-robin
// to the 4 waves (independent loads in flight together), then the 4 partial sums added in order
__global__ __launch_bounds__(256) void type_embed_grad_finish_kernel(const float* __restrict__ part, int nblocks, int d,
                                                                     float* __restrict__ demb) {
    __shared__ float red[4][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int o = blockIdx.x * 64 + tx;
    const int oc = min(o, 2 * d - 1);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int rb = ty;
    for (; rb + 12 < nblocks; rb += 16) {
        s0 += part[(int64_t)(rb + 0) * 2 * d + oc];
        s1 += part[(int64_t)(rb + 4) * 2 * d + oc];
        s2 += part[(int64_t)(rb + 8) * 2 * d + oc];
        s3 += part[(int64_t)(rb + 12) * 2 * d + oc];
    }
    for (; rb < nblocks; rb += 4) s0 += part[(int64_t)rb * 2 * d + oc];
    red[ty][tx] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (ty == 0 && o < 2 * d) demb[o] = ((red[0][tx] + red[1][tx]) + red[2][tx]) + red[3][tx];
}

static inline int te_row_blocks(int64_t N, int d, int vecn) {
    const int ctiles = cdiv(d, 64 * vecn);
    int64_t want = 1024 / ctiles;                      // ~4 workgroups per CU in total
    if (want < 1) want = 1;
    int64_t rpb = (N + want - 1) / want;
    if (rpb < 16) rpb = 16;
    return (int)((N + rpb - 1) / rpb);
}

}  // namespace star

extern "C" int star_type_embed_add(const void* x, const int64_t* table, const void* emb, int dtype, int64_t N, int d,
                                   void* out, void* stream) {
    using namespace star;
    STAR_REQUIRE(x && table && emb && out, "star_type_embed_add: null pointer");
    STAR_REQUIRE(dtype == STAR_DT_BF16 || dtype == STAR_DT_F32, "star_type_embed_add: dtype %d not bf16/f32", dtype);
    const int vecn = dtype == STAR_DT_BF16 ? 8 : 4;
    STAR_REQUIRE(N >= 1 && d >= vecn && d % vecn == 0, "star_type_embed_add: bad shape N=%lld d=%d (d must be a multiple of %d)",
                 (long long)N, d, vecn);
    const int64_t nvec = N * (d / vecn);
    STAR_REQUIRE(nvec / 256 < (1ll << 31) - 1, "star_type_embed_add: too many elements");
    if (dtype == STAR_DT_BF16)
        hipLaunchKernelGGL(type_embed_add_kernel<__bf16>, dim3(cdiv(nvec, 256)), dim3(256), 0, (hipStream_t)stream,
                           (const __bf16*)x, table, (const __bf16*)emb, nvec, d / vecn, (__bf16*)out);
    else
        hipLaunchKernelGGL(type_embed_add_kernel<float>, dim3(cdiv(nvec, 256)), dim3(256), 0, (hipStream_t)stream,
                           (const float*)x, table, (const float*)emb, nvec, d / vecn, (float*)out);
    STAR_LAUNCH_CHECK("star_type_embed_add");
    return STAR_OK;
}

extern "C" size_t star_type_embed_grad_workspace(int64_t N, int d, int dtype) {
    using namespace star;
    if (N <= 0 || d <= 0) return 0;
    return (size_t)te_row_blocks(N, d, dtype == STAR_DT_BF16 ? 8 : 4) * 2 * (size_t)d * sizeof(float);
}

extern "C" int star_type_embed_grad(const void* dout, const int64_t* table, int dtype, int64_t N, int d, float* demb,
                                    void* workspace, size_t workspace_bytes, void* stream) {
    using namespace star;
    STAR_REQUIRE(dout && table && demb && workspace, "star_type_embed_grad: null pointer");
    STAR_REQUIRE(dtype == STAR_DT_BF16 || dtype == STAR_DT_F32, "star_type_embed_grad: dtype %d not bf16/f32", dtype);
    const int vecn = dtype == STAR_DT_BF16 ? 8 : 4;
    STAR_REQUIRE(N >= 1 && d >= vecn && d % vecn == 0, "star_type_embed_grad: bad shape N=%lld d=%d", (long long)N, d);
    if (workspace_bytes < star_type_embed_grad_workspace(N, d, dtype)) {
        set_error("star_type_embed_grad: workspace %zu < %zu bytes", workspace_bytes, star_type_embed_grad_workspace(N, d, dtype));
        return STAR_ERR_WORKSPACE;
    }
    const int nblocks = te_row_blocks(N, d, vecn);
    const int rpb = (int)((N + nblocks - 1) / nblocks);
    STAR_REQUIRE(nblocks <= 65535, "star_type_embed_grad: too many row blocks");
    float* part = (float*)workspace;
    const dim3 grid(cdiv(d, 64 * vecn), nblocks);
    if (dtype == STAR_DT_BF16)
        hipLaunchKernelGGL(type_embed_grad_kernel<__bf16>, grid, dim3(256), 0, (hipStream_t)stream, (const __bf16*)dout,
                           table, N, d, rpb, part);
    else
        hipLaunchKernelGGL(type_embed_grad_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)dout,
                           table, N, d, rpb, part);
    hipLaunchKernelGGL(type_embed_grad_finish_kernel, dim3(cdiv(2 * d, 64)), dim3(256), 0, (hipStream_t)stream,
                       (const float*)part, nblocks, d, demb);
    STAR_LAUNCH_CHECK("star_type_embed_grad");
    return STAR_OK;
}
