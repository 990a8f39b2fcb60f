// Row gather into the padded nugget memory and its backward.
//
// Forward  (src/models/w2v_asr.py:566-567, w2v_simul.py:427,446-447):
//     dst[b,j,:] = src[b, idx[b,j], :]         idx < 0 wraps like Python (-1 = last frame)
//   one wave per output row, 16 B per lane per step (coalesced 1 KiB wave accesses).
//   HBM bytes: read + write 2 * row_bytes * B*K.
// Backward (autograd of the advanced index = index_put accumulate):
//     dsrc[b,t,:] = sum_{j : idx[b,j] == t} ddst[b,j,:]
//   destination-centric: one workgroup per (b, 64-frame tile) inverts idx for its tile in LDS, then
//   every destination row is written exactly once (copy, zero, or ordered fp32 sum for duplicates).
//   No atomics, no separate zero-fill pass, bitwise reproducible.
//   HBM bytes: read row_bytes * B*K, write row_bytes * B*T.
#include "common.h"

namespace star {

// Index normalisation shared by the forward and every backward path: Python-style wrap of negative indices, then a
// CLAMP into [0, T) (torch would raise IndexError for an index outside [-T, T); a kernel cannot, so both directions
// agree on the same frame instead: the gradient of a clamped slot lands on the frame the forward read).
__device__ __forceinline__ int64_t gather_frame(int64_t t, int T) {
    if (t < 0) t += T;
    if (t < 0) t = 0;
    if (t >= T) t = T - 1;
    return t;
}

template <typename V>
__global__ __launch_bounds__(256) void gather_rows_fwd_kernel(const V* __restrict__ src,
                                                              const int64_t* __restrict__ idx,
                                                              V* __restrict__ dst, int T, int K,
                                                              int64_t rows_total, int vec_per_row) {
    const int lane = threadIdx.x & 63;
    int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows_total) return;
    int64_t b = row / K;
    const int64_t t = gather_frame(idx[row], T);
    const V* s = src + ((size_t)b * T + t) * vec_per_row;
    V* d = dst + (size_t)row * vec_per_row;
    for (int c = lane; c < vec_per_row; c += 64) d[c] = s[c];
}

constexpr int GB_TILE = 64;  // destination frames per workgroup

template <typename T, int N>
struct alignas(sizeof(T) * N) Pack {
    T v[N];
};

// T_ELEM: __bf16 or float; rows are d elements, moved VEC elements per lane per step
// (VEC = 4 when d % 4 == 0, else 1 -- the scalar score gather of w2v_simul.py:427 has d = 1).
constexpr int GB_DUP = 16;     // source rows of a duplicated frame that a wave keeps in flight

template <typename T_ELEM, int VEC>
__global__ __launch_bounds__(256) void gather_rows_bwd_kernel(const T_ELEM* __restrict__ ddst,
                                                              const int64_t* __restrict__ idx,
                                                              T_ELEM* __restrict__ dsrc, int T, int K, int d) {
    __shared__ int first_j[GB_TILE];
    __shared__ int count[GB_TILE];
    __shared__ int copy_list[GB_TILE];
    __shared__ int ncopy_s;
    static_assert(GB_TILE == 64, "one lane per row of the tile builds the copy list");
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int b = blockIdx.y;
    const int t0 = blockIdx.x * GB_TILE;
    if (tid < GB_TILE) {
        first_j[tid] = 0x7FFFFFFF;
        count[tid] = 0;
    }
    __syncthreads();
    const int64_t* irow = idx + (size_t)b * K;
    for (int j = tid; j < K; j += 256) {
        const int64_t t = gather_frame(irow[j], T);
        int r = (int)(t - t0);
        if (r >= 0 && r < GB_TILE) {
            atomicAdd(&count[r], 1);
            atomicMin(&first_j[r], j);
        }
    }
    __syncthreads();
    typedef Pack<T_ELEM, VEC> vec4;
    const int nvec = d / VEC;
    // Rows of up to GB_FV vectors per lane (2 KB at 16-byte vectors): the tile's single-source rows are listed first, a
    // wave requests four of them (whole rows, all loads before anything else), writes its zero rows while they are in
    // flight and stores the copies last -- row by row in program order every copied row cost the wave one exposed
    // round trip to HBM.
    // The loads are inline asm: written as C++ the compiler sinks each one into the (conditional) block of its store,
    // behind the zero stores, with a vmcnt(0) of its own.  Untracked, so the wave waits by hand before the copies.
    constexpr int GB_FV = 2;
    const bool fast = sizeof(vec4) == 16 && nvec <= 64 * GB_FV;
    // Frames with several sources (in this pipeline: frame T-1 of an utterance, which the -1 pad slots address --
    // 90 of 116 slots at MuST-C-shaped lengths) come first and are summed by the WHOLE workgroup, every wave a quarter
    // of the columns: fp32 sum in ascending j per element (the documented order), GB_DUP source rows in flight per wave.
    // One wave per row with four rows in flight made this row the kernel's longest chain (23 HBM round trips).
    const int nvq = (nvec + 3) >> 2;                   // vectors per wave
    const bool coop = nvq <= 64;
    if (coop) {
        const int vq = wid * nvq + lane;
        const bool mine = lane < nvq && vq < nvec;
        const int vc = min(vq, nvec - 1);
        unsigned long long dup_rows = __ballot(count[lane] >= 2 && t0 + lane < T);      // one lane per row of the tile
        while (dup_rows) {
            const int r = __builtin_ctzll(dup_rows);
            dup_rows &= dup_rows - 1;
            const int t = t0 + r;
            float acc[VEC];
#pragma unroll
            for (int q = 0; q < VEC; ++q) acc[q] = 0.f;
            const int jf = first_j[r];
            for (int j0 = jf & ~63; j0 < K; j0 += 64) {
                const int j = j0 + lane;
                bool m = false;
                if (j < K && j >= jf) m = (gather_frame(irow[j], T) == t);
                unsigned long long bal = __ballot(m);
                while (bal) {
                    int ks[GB_DUP];
                    const int kfirst = __builtin_ctzll(bal);
#pragma unroll
                    for (int u = 0; u < GB_DUP; ++u) {
                        ks[u] = bal ? __builtin_ctzll(bal) : -1;
                        bal &= bal - 1;          // (0 & anything stays 0)
                    }
                    // every load is unconditional (slots / vectors that do not exist re-read a valid address and are
                    // dropped at the add): a branch around a load makes the compiler drain vmcnt(0) per element
                    vec4 x[GB_DUP];
#pragma unroll
                    for (int u = 0; u < GB_DUP; ++u) {
                        const int ku = ks[u] >= 0 ? ks[u] : kfirst;
                        x[u] = ((const vec4*)(ddst + ((size_t)b * K + j0 + ku) * d))[vc];
                    }
#pragma unroll
                    for (int u = 0; u < GB_DUP; ++u)
#pragma unroll
                        for (int q = 0; q < VEC; ++q) {
                            const float xv = (float)x[u].v[q];
                            acc[q] += (ks[u] >= 0) ? xv : 0.0f;      // a missing slot adds +0
                        }
                }
            }
            if (mine) {
                vec4 o;
#pragma unroll
                for (int q = 0; q < VEC; ++q) o.v[q] = (T_ELEM)acc[q];
                ((vec4*)(dsrc + ((size_t)b * T + t) * d))[vq] = o;
            }
        }
    }
    // the other rows: a wave per row
    for (int r = wid; r < GB_TILE; r += 4) {
        int t = t0 + r;
        if (t >= T) break;
        vec4* out = (vec4*)(dsrc + ((size_t)b * T + t) * d);
        int c = count[r];
        if (fast && c <= 1) continue;                  // done below
        if (coop && c >= 2) continue;                  // done above
        if (c == 0) {
            vec4 z;
#pragma unroll
            for (int q = 0; q < VEC; ++q) z.v[q] = (T_ELEM)0.0f;
            for (int v = lane; v < nvec; v += 64) out[v] = z;
        } else if (c == 1) {
            const vec4* in = (const vec4*)(ddst + ((size_t)b * K + first_j[r]) * d);
            for (int v = lane; v < nvec; v += 64) out[v] = in[v];
        } else {
            // very wide rows: one vector at a time
            for (int v = lane; v < nvec; v += 64) {
                float acc[VEC];
#pragma unroll
                for (int q = 0; q < VEC; ++q) acc[q] = 0.f;
                for (int j = first_j[r]; j < K; ++j) {
                    if (gather_frame(irow[j], T) != t) continue;
                    vec4 x = ((const vec4*)(ddst + ((size_t)b * K + j) * d))[v];
#pragma unroll
                    for (int q = 0; q < VEC; ++q) acc[q] += (float)x.v[q];
                }
                vec4 o;
#pragma unroll
                for (int q = 0; q < VEC; ++q) o.v[q] = (T_ELEM)acc[q];
                out[v] = o;
            }
        }
    }
    if (fast) {
        if (wid == 0) {
            const bool one = count[lane] == 1 && t0 + lane < T;
            const unsigned long long bal = __ballot(one);
            if (one) copy_list[__builtin_popcountll(bal & ((1ull << lane) - 1ull))] = lane;
            if (lane == 0) ncopy_s = __builtin_popcountll(bal);
        }
        __syncthreads();
        const int ncopy = ncopy_s;
        const u32x4 z = {0u, 0u, 0u, 0u};
        for (int e0 = 0; e0 < max(ncopy, 1); e0 += 16) {
            u32x4 x[4][GB_FV];
            int rr[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = e0 + wid + 4 * u;
                // entries past the list re-read a valid row (the loads carry no branch); their store is skipped
                const int r = ncopy > 0 ? copy_list[min(e, ncopy - 1)] : 0;
                const int j = ncopy > 0 ? first_j[r] : 0;
                rr[u] = e < ncopy ? r : -1;
                const u32x4* in = (const u32x4*)(ddst + ((size_t)b * K + j) * d);
#pragma unroll
                for (int i = 0; i < GB_FV; ++i) {
                    const u32x4* pv = in + min(lane + 64 * i, nvec - 1);
                    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(x[u][i]) : "v"(pv) : "memory");
                }
            }
            if (e0 == 0) {
                for (int r = wid; r < GB_TILE; r += 4) {
                    const int t = t0 + r;
                    if (t >= T) break;
                    if (count[r] != 0) continue;
                    u32x4* out = (u32x4*)(dsrc + ((size_t)b * T + t) * d);
#pragma unroll
                    for (int i = 0; i < GB_FV; ++i)
                        if (lane + 64 * i < nvec) out[lane + 64 * i] = z;
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (rr[u] < 0) continue;
                u32x4* out = (u32x4*)(dsrc + ((size_t)b * T + t0 + rr[u]) * d);
#pragma unroll
                for (int i = 0; i < GB_FV; ++i)
                    if (lane + 64 * i < nvec) out[lane + 64 * i] = x[u][i];
            }
        }
    }
}

}  // namespace star

extern "C" int star_gather_rows_fwd(const void* src, const int64_t* idx, void* dst, int B, int T, int K,
                                    int64_t row_bytes, void* stream) {
    using namespace star;
    STAR_REQUIRE(src && idx && dst, "star_gather_rows_fwd: null pointer");
    STAR_REQUIRE(B >= 1 && T >= 1 && K >= 1 && row_bytes >= 2 && row_bytes % 2 == 0,
                 "star_gather_rows_fwd: bad shape B=%d T=%d K=%d row_bytes=%lld", B, T, K, (long long)row_bytes);
    int64_t rows = (int64_t)B * K;
    dim3 grid(cdiv(rows, 4)), block(256);
    bool a16 = (row_bytes % 16 == 0) && (((uintptr_t)src | (uintptr_t)dst) % 16 == 0);
    if (row_bytes % 4 != 0 || (((uintptr_t)src | (uintptr_t)dst) % 4 != 0))
        // 2-byte granules: the bf16 score gather (d = 1, w2v_simul.py:427 under bf16) and odd bf16 row lengths
        hipLaunchKernelGGL(gather_rows_fwd_kernel<uint16_t>, grid, block, 0, (hipStream_t)stream,
                           (const uint16_t*)src, idx, (uint16_t*)dst, T, K, rows, (int)(row_bytes / 2));
    else if (a16)
        hipLaunchKernelGGL(gather_rows_fwd_kernel<u32x4>, grid, block, 0, (hipStream_t)stream, (const u32x4*)src,
                           idx, (u32x4*)dst, T, K, rows, (int)(row_bytes / 16));
    else
        hipLaunchKernelGGL(gather_rows_fwd_kernel<uint32_t>, grid, block, 0, (hipStream_t)stream,
                           (const uint32_t*)src, idx, (uint32_t*)dst, T, K, rows, (int)(row_bytes / 4));
    STAR_LAUNCH_CHECK("star_gather_rows_fwd");
    return STAR_OK;
}

// d = 1 (the score gather of w2v_asr.py:244 / w2v_simul.py:427): one workgroup per utterance builds the whole dense
// row in LDS.  Targets hit once take their value directly; a target hit several times (pad slots all address frame
// T-1) is summed by the thread of its first slot in ascending slot order (fp32, one rounding at the end) -- the same
// result as the tiled kernel above, which spends a wave per frame on one element when d = 1.
constexpr int GS_MAX = 6144;       // T and K limit of the LDS-resident variant (120 KB of LDS at the limit)
template <typename T_ELEM>
__global__ __launch_bounds__(256) void scatter_scalar_bwd_kernel(const T_ELEM* __restrict__ ddst,
                                                                 const int64_t* __restrict__ idx,
                                                                 T_ELEM* __restrict__ dsrc, int T, int K) {
    extern __shared__ int gs_smem[];
    int* cnt = gs_smem;                        // [T]
    int* first = cnt + T;                      // [T]
    float* acc = (float*)(first + T);          // [T]
    int* tj = (int*)(acc + T);                 // [K] target frame of slot j (-1: out of range)
    float* gj = (float*)(tj + K);              // [K]
    const int tid = threadIdx.x, b = blockIdx.x;
    for (int t = tid; t < T; t += 256) {
        cnt[t] = 0;
        first[t] = 0x7FFFFFFF;
        acc[t] = 0.f;
    }
    for (int j = tid; j < K; j += 256) {
        tj[j] = (int)star::gather_frame(idx[(size_t)b * K + j], T);
        gj[j] = (float)ddst[(size_t)b * K + j];
    }
    __syncthreads();
    for (int j = tid; j < K; j += 256) {
        const int t = tj[j];
        if (t >= 0) {
            atomicAdd(&cnt[t], 1);
            atomicMin(&first[t], j);
        }
    }
    __syncthreads();
    for (int j = tid; j < K; j += 256) {
        const int t = tj[j];
        if (t < 0) continue;
        if (cnt[t] == 1) {
            acc[t] = gj[j];
        } else if (first[t] == j) {
            float sum = 0.f;
            // ascending slot order (the documented fp32 order); unrolled so that the LDS reads of eight slots are in
            // flight together -- one slot per round trip made 90 pad slots on frame T-1 cost 7 us
#pragma unroll 8
            for (int jj = j; jj < K; ++jj) sum += (tj[jj] == t) ? gj[jj] : 0.f;
            acc[t] = sum;
        }
    }
    __syncthreads();
    for (int t = tid; t < T; t += 256) dsrc[(size_t)b * T + t] = (T_ELEM)acc[t];
}

extern "C" int star_gather_rows_bwd(const void* ddst, const int64_t* idx, void* dsrc, int B, int T, int K, int d,
                                    int dtype, void* stream) {
    using namespace star;
    STAR_REQUIRE(ddst && idx && dsrc, "star_gather_rows_bwd: null pointer");
    STAR_REQUIRE(B >= 1 && T >= 1 && K >= 1 && d >= 1, "star_gather_rows_bwd: bad shape B=%d T=%d K=%d d=%d", B, T, K, d);
    STAR_REQUIRE(dtype == STAR_DT_BF16 || dtype == STAR_DT_F32, "star_gather_rows_bwd: bad dtype %d", dtype);
    STAR_REQUIRE(B <= 65535, "star_gather_rows_bwd: B=%d exceeds grid.y", B);
    hipStream_t st = (hipStream_t)stream;
    if (d == 1 && T <= GS_MAX && K <= GS_MAX) {
        const size_t smem = ((size_t)3 * T + (size_t)2 * K) * 4;
        if (smem > 48 * 1024) {
            static DevOnce once[2];
            const int w = dtype == STAR_DT_BF16 ? 0 : 1;
            const void* fn = w == 0 ? (const void*)scatter_scalar_bwd_kernel<__bf16>
                                    : (const void*)scatter_scalar_bwd_kernel<float>;
            if (!reserve_lds_once(once[w], fn, (3 * GS_MAX + 2 * GS_MAX) * 4)) {
                set_error("star_gather_rows_bwd: cannot reserve LDS");
                return STAR_ERR_LAUNCH;
            }
        }
        if (dtype == STAR_DT_BF16)
            hipLaunchKernelGGL(scatter_scalar_bwd_kernel<__bf16>, dim3(B), dim3(256), smem, st, (const __bf16*)ddst, idx,
                               (__bf16*)dsrc, T, K);
        else
            hipLaunchKernelGGL(scatter_scalar_bwd_kernel<float>, dim3(B), dim3(256), smem, st, (const float*)ddst, idx,
                               (float*)dsrc, T, K);
        STAR_LAUNCH_CHECK("star_gather_rows_bwd");
        return STAR_OK;
    }
    dim3 grid(cdiv(T, GB_TILE), B), block(256);
    const int esz = dtype == STAR_DT_BF16 ? 2 : 4;
    const bool v4 = (d % 4 == 0) && (((uintptr_t)ddst | (uintptr_t)dsrc) % (4 * esz) == 0);
    const bool v8 = dtype == STAR_DT_BF16 && (d % 8 == 0) && (((uintptr_t)ddst | (uintptr_t)dsrc) % 16 == 0);
    if (dtype == STAR_DT_BF16) {
        if (v8)
            hipLaunchKernelGGL((gather_rows_bwd_kernel<__bf16, 8>), grid, block, 0, st, (const __bf16*)ddst, idx,
                               (__bf16*)dsrc, T, K, d);
        else if (v4)
            hipLaunchKernelGGL((gather_rows_bwd_kernel<__bf16, 4>), grid, block, 0, st, (const __bf16*)ddst, idx,
                               (__bf16*)dsrc, T, K, d);
        else
            hipLaunchKernelGGL((gather_rows_bwd_kernel<__bf16, 1>), grid, block, 0, st, (const __bf16*)ddst, idx,
                               (__bf16*)dsrc, T, K, d);
    } else {
        if (v4)
            hipLaunchKernelGGL((gather_rows_bwd_kernel<float, 4>), grid, block, 0, st, (const float*)ddst, idx,
                               (float*)dsrc, T, K, d);
        else
            hipLaunchKernelGGL((gather_rows_bwd_kernel<float, 1>), grid, block, 0, st, (const float*)ddst, idx,
                               (float*)dsrc, T, K, d);
    }
    STAR_LAUNCH_CHECK("star_gather_rows_bwd");
    return STAR_OK;
}
