// CIF (continuous integrate-and-fire) weighted aggregation -- SURVEY.md section 8f item 2.
//
// Reference: the frame loop of W2VSimulModel.lm_cif_training (src/models/w2v_simul.py:558-606; the same loop in
// W2VASRModel.lm_cif_training, w2v_asr.py:681-722): T Python iterations, each a handful of [B] / [B,H] launches,
// then a per-utterance index_select + pad.  Restated:
//     dc = 1 - I;  I += a_t;  fire = I >= thr;  I -= 1 where fired;  cur = fire ? dc : a_t;  rem = a_t - cur
//     frame += cur * s_t;  emit frame at a fire;  frame = rem * s_t after a fire
// (no forced fire at the last valid frame; the partial segment after the last fire is dropped).
//
// Here: (1) cif_weights_kernel -- the scalar recurrence, one lane per utterance on rows resident in LDS (as in the
// nugget fire scan), emits cur / rem / segment id per frame and the fire positions; (2) cif_aggregate_fwd_kernel -- one workgroup per emitted row walks its frames once, fp32 accumulate in
// the reference's order (mul then add, no contraction); (3) backward: one wave per frame forms dS_t and the two
// dot products <s_t, dOut> that are d cur_t / d rem_t, then a per-utterance reverse scan turns them into d alphas
// (cur_t of a firing frame is 1 - sum of ALL earlier alphas + const, so its gradient reaches every earlier frame).
// HBM traffic: forward reads S once (+ the firing rows twice), backward reads S once and writes dS once.
#include "common.h"

namespace star {

__host__ __device__ static inline int cw_pitch(int T) { return ((T + 15) & ~15) + 4; }
static inline size_t cw_row_bytes(int T) { return (size_t)2 * cw_pitch(T) * 4; }
static inline int cw_rows_per_wg(int B, int T) {
    int r = 4;      // one utterance per workgroup unless the batch is huge (see scan_rows_per_wg in cif.hip)
    while (r > 1 && ((size_t)r * cw_row_bytes(T) > 64 * 1024 || (B + r - 1) / r < 2048)) r >>= 1;
    return r;
}

// A workgroup owns R utterances.  (1) all 256 threads bring their alpha rows into LDS twice (coalesced): one copy is
// kept, the other is the work row of the recurrence; (2) lane r runs cif_serial_chain() on work row r (4 instructions
// per frame), which leaves the pre-fire integrals I_t there; (3) all threads turn {alpha, I} into fires, the integral
// before every frame, the fire positions (order-preserving compaction by an exclusive scan over contiguous chunks)
// and the per-frame records {cur, rem, segment/fire bits} in parallel.
__global__ __launch_bounds__(256) void cif_weights_kernel(const float* __restrict__ alphas, float threshold, int B, int T,
                                                         int cap, int R, f32x4* __restrict__ w_out,
                                                         int64_t* __restrict__ idx_out, uint8_t* __restrict__ mask_out,
                                                         int32_t* __restrict__ counts, int32_t* __restrict__ max_count) {
    extern __shared__ __attribute__((aligned(16))) float cw_smem[];
    __shared__ int wtot[4], slow[4];
    const int pitch = cw_pitch(T), T16 = (T + 15) & ~15, cpitch = pitch >> 4;
    float* a_rows = cw_smem;                                  // [R][pitch] alphas
    float* i_rows = a_rows + R * pitch;                       // [R][pitch] alphas -> integral after the add
    float* carries = i_rows + R * pitch;                      // [R][pitch / 16] carry into every 16-frame block
    const int tid = threadIdx.x;
    const int b0 = blockIdx.x * R;
    const int nrows = min(R, B - b0);
    if (tid < 4) slow[tid] = (threshold != 1.0f) ? 1 : 0;
    __syncthreads();
    for (int r = 0; r < nrows; ++r) {
        bool odd = false;                                     // a weight outside {0} u [FLT_MIN, 1) (NaN included)
        for (int t = tid; t < T16; t += 256) {
            const float a = (t < T) ? alphas[(size_t)(b0 + r) * T + t] : 0.0f;
            odd |= !(a == 0.0f || (a >= 1.17549435e-38f && a < 1.0f));
            a_rows[r * pitch + t] = a;
            i_rows[r * pitch + t] = a;
        }
        if (odd) slow[r] = 1;
    }
    __syncthreads();
    // threshold 1 and every weight in {0} u [FLT_MIN, 1): the two-instruction chain (common.h) with only the block
    // carries kept by the serial lane; the integrals are rebuilt by all threads (bit-identical to the generic chain)
    if (tid < nrows) {
        if (!slow[tid]) cif_serial_carries_unit(i_rows + tid * pitch, carries + tid * cpitch, T16 >> 4, 0.0f);
        else cif_serial_chain<false>(i_rows + tid * pitch, nullptr, threshold, T16 >> 4);
    }
    __syncthreads();
    for (int r = 0; r < nrows; ++r) {
        if (slow[r]) continue;
        for (int blk = tid; blk < (T16 >> 4); blk += 256) cif_expand_block_unit(i_rows + r * pitch, carries + r * cpitch, blk);
    }
    __syncthreads();
    const int C = (T + 255) / 256;
    for (int r = 0; r < nrows; ++r) {
        const int b = b0 + r;
        const float* ar = a_rows + r * pitch;
        const float* ir = i_rows + r * pitch;
        const size_t ob = (size_t)b * T;
        const int lo = min(tid * C, T), hi = min(lo + C, T);
        int cnt = 0;
        for (int t = lo; t < hi; ++t) cnt += (ir[t] >= threshold) ? 1 : 0;
        int nfire;
        int before = block_excl_scan_256(cnt, wtot, nfire);
        for (int t = lo; t < hi; ++t) {
            const float it = ir[t];
            const bool fire = it >= threshold;
            // the integral the frame started from: the previous frame's, less 1 if that one fired
            float prev = 0.0f;
            if (t > 0) {
                const float ip = ir[t - 1];
                prev = (ip >= threshold) ? __fsub_rn(ip, 1.0f) : ip;
            }
            const float a = ar[t];
            const float c = fire ? __fsub_rn(1.0f, prev) : a;
            // one 16-byte record per frame: {cur, rem, (fires before t) << 1 | fired, 0}
            w_out[ob + t] = (f32x4){c, __fsub_rn(a, c), __int_as_float((before << 1) | (fire ? 1 : 0)), 0.f};
            if (fire) {
                if (before < cap) {
                    idx_out[(size_t)b * cap + before] = t;
                    if (mask_out) mask_out[(size_t)b * cap + before] = 1;
                }
                ++before;
            }
        }
        if (tid == 0) {
            counts[b] = nfire;
            if (max_count) atomicMax(max_count, nfire);
        }
        for (int j = nfire + tid; j < cap; j += 256) {
            idx_out[(size_t)b * cap + j] = -1;
            if (mask_out) mask_out[(size_t)b * cap + j] = 0;
        }
    }
}

template <typename T, int N> struct __attribute__((aligned(sizeof(T) * N))) Pack { T v[N]; };

template <typename T> __device__ __forceinline__ float ld1(const T* p);
template <> __device__ __forceinline__ float ld1<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ld1<__bf16>(const __bf16* p) { return (float)*p; }
template <typename T> __device__ __forceinline__ void st1(T* p, float v);
template <> __device__ __forceinline__ void st1<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void st1<__bf16>(__bf16* p, float v) { *p = (__bf16)v; }

// out[b, j, :] = rem[f_{j-1}] * s[f_{j-1}] + sum_{t = f_{j-1}+1 .. f_j} cur[t] * s[t]   (f_{-1}: none, start at frame 0)
// One workgroup per (b, j); thread = VEC consecutive channels; frames in ascending order, mul and add rounded
// separately like the reference's `frame += cur[:, None] * s_t`.
template <typename T, int VEC>
__global__ __launch_bounds__(128) void cif_aggregate_fwd_kernel(const T* __restrict__ s, const f32x4* __restrict__ w,
                                                               const int64_t* __restrict__ idx,
                                                               const int32_t* __restrict__ counts, int Tn, int cap, int d,
                                                               T* __restrict__ out) {
    const int j = blockIdx.x, b = blockIdx.y;
    const int cnt = min(counts[b], cap);
    T* orow = out + ((size_t)b * cap + j) * d;
    if (j >= cnt) {
        for (int c = threadIdx.x * VEC; c < d; c += 128 * VEC)
#pragma unroll
            for (int v = 0; v < VEC; ++v) st1<T>(orow + c + v, 0.0f);
        return;
    }
    const int t_end = (int)idx[(size_t)b * cap + j];
    const int t_prev = (j > 0) ? (int)idx[(size_t)b * cap + j - 1] : -1;
    const T* sb = s + (size_t)b * Tn * d;
    const f32x4* wb = w + (size_t)b * Tn;
    for (int c = threadIdx.x * VEC; c < d; c += 128 * VEC) {
        float acc[VEC];
        if (t_prev >= 0) {
            const float w = wb[t_prev][1];
#pragma unroll
            for (int v = 0; v < VEC; ++v) acc[v] = __fmul_rn(w, ld1<T>(sb + (size_t)t_prev * d + c + v));
        } else {
#pragma unroll
            for (int v = 0; v < VEC; ++v) acc[v] = 0.0f;
        }
        for (int t = t_prev + 1; t <= t_end; ++t) {
            const float w = wb[t][0];
#pragma unroll
            for (int v = 0; v < VEC; ++v) acc[v] = __fadd_rn(acc[v], __fmul_rn(w, ld1<T>(sb + (size_t)t * d + c + v)));
        }
#pragma unroll
        for (int v = 0; v < VEC; ++v) st1<T>(orow + c + v, acc[v]);
    }
}

// One wave per frame (b, t):  dS_t = cur_t * g1 + rem_t * g2,  dcur_t = <s_t, g1>,  drem_t = <s_t, g2>
// with g1 = dOut[b, seg_t] (if that row was emitted) and g2 = dOut[b, seg_t + 1] (if t fired and that row exists).
// Loads are unconditional vector loads (rows that do not exist re-read row 0 and get weight 0).
template <typename T, int VEC>
__global__ __launch_bounds__(256) void cif_aggregate_bwd_kernel(const T* __restrict__ dout, const T* __restrict__ s,
                                                               const f32x4* __restrict__ w,
                                                               const int32_t* __restrict__ counts, int B, int Tn, int cap,
                                                               int d, T* __restrict__ ds, float* __restrict__ dcur,
                                                               float* __restrict__ drem) {
    typedef Pack<T, VEC> vec;
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= (int64_t)B * Tn) return;
    const int b = (int)(row / Tn);
    const int cnt = min(counts[b], cap);
    const f32x4 wr = w[row];
    const int enc = __float_as_int(wr[2]);
    const int j = enc >> 1;
    const bool fire = enc & 1;
    const bool has1 = j < cnt, has2 = fire && (j + 1 < cnt);
    const float c = has1 ? wr[0] : 0.f, r = has2 ? wr[1] : 0.f;
    const vec* srow = (const vec*)(s + row * d);
    const vec* g1 = (const vec*)(dout + ((size_t)b * cap + (has1 ? j : 0)) * d);
    const vec* g2 = (const vec*)(dout + ((size_t)b * cap + (has2 ? j + 1 : 0)) * d);
    vec* drow = (vec*)(ds + row * d);
    float a1 = 0.f, a2 = 0.f;
    const int nvec = d / VEC;
    for (int v = lane; v < nvec; v += 64) {
        const vec sv = srow[v], v1 = g1[v], v2 = g2[v];
        vec o;
#pragma unroll
        for (int q = 0; q < VEC; ++q) {
            const float sf = (float)sv.v[q];
            const float f1 = has1 ? (float)v1.v[q] : 0.f, f2 = has2 ? (float)v2.v[q] : 0.f;   // selects, not branches
            a1 += sf * f1;
            a2 += sf * f2;
            o.v[q] = (T)(c * f1 + r * f2);
        }
        drow[v] = o;
    }
    a1 = wave_sum(a1);
    a2 = wave_sum(a2);
    if (lane == 0) {
        dcur[row] = a1;
        drem[row] = a2;
    }
}

// d alpha_u = (fire_u ? drem_u : dcur_u) + sum_{t > u, fire_t} (drem_t - dcur_t): a suffix sum per utterance.  One
// 256-thread workgroup per utterance, every thread owns a contiguous chunk of frames (all its loads in flight at once),
// the chunk totals go through an exclusive suffix scan over the workgroup (fixed order: reproducible).
__global__ __launch_bounds__(256) void cif_dalpha_kernel(const float* __restrict__ dcur, const float* __restrict__ drem,
                                                        const f32x4* __restrict__ w, int Tn,
                                                        float* __restrict__ dalphas) {
    __shared__ float wtot[4];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int C = (Tn + 255) / 256;
    const int lo = min(tid * C, Tn), hi = min(lo + C, Tn);
    const size_t base = (size_t)b * Tn;
    // pass 1: the chunk's total of (drem - dcur) over firing frames
    float tot = 0.f;
    for (int t = hi - 1; t >= lo; --t) {
        const bool fire = __float_as_int(w[base + t][2]) & 1;
        tot += fire ? (drem[base + t] - dcur[base + t]) : 0.f;
    }
    // exclusive suffix sum of the chunk totals: lanes, then waves (later chunks = higher thread ids)
    float incl = tot;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const float n = __shfl_down(incl, off, 64);
        if (lane + off < 64) incl += n;
    }
    if (lane == 0) wtot[wid] = incl;
    __syncthreads();
    float after = incl - tot;
    for (int x = 3; x > wid; --x) after += wtot[x];
    // pass 2: walk the chunk backwards (the values are in L1/L2 now)
    float run = after;
    for (int t = hi - 1; t >= lo; --t) {
        const bool fire = __float_as_int(w[base + t][2]) & 1;
        const float dc = dcur[base + t], dr = drem[base + t];
        dalphas[base + t] = (fire ? dr : dc) + run;
        run += fire ? (dr - dc) : 0.f;
    }
}

}  // namespace star

extern "C" int star_cif_weights(const float* alphas, float threshold, int B, int T, int cap, float* weights,
                                int64_t* idx_out, uint8_t* mask_out, int32_t* counts, int32_t* max_count,
                                void* stream) {
    using namespace star;
    STAR_REQUIRE(alphas && weights && idx_out && counts, "star_cif_weights: null pointer");
    STAR_REQUIRE((uintptr_t)weights % 16 == 0, "star_cif_weights: weights must be 16-byte aligned");
    STAR_REQUIRE(B >= 1 && T >= 1 && cap >= 1, "star_cif_weights: bad shape B=%d T=%d cap=%d", B, T, cap);
    if (max_count) {
        hipError_t e = hipMemsetAsync(max_count, 0, sizeof(int32_t), (hipStream_t)stream);
        if (e != hipSuccess) {
            set_error("star_cif_weights: memset failed: %s", hipGetErrorString(e));
            return STAR_ERR_LAUNCH;
        }
    }
    STAR_REQUIRE(T <= 16384, "star_cif_weights: T=%d exceeds the 16384-frame LDS row limit", T);
    const int R = cw_rows_per_wg(B, T);
    const size_t smem = (size_t)R * cw_row_bytes(T) + (size_t)R * (cw_pitch(T) >> 4) * 4;
    STAR_REQUIRE(T < 32768, "star_cif_weights: T too large");
    if (smem > 48 * 1024) {
        static DevOnce once;
        if (!reserve_lds_once(once, (const void*)cif_weights_kernel, 140 * 1024)) {
            set_error("star_cif_weights: cannot reserve LDS");
            return STAR_ERR_LAUNCH;
        }
    }
    hipLaunchKernelGGL(cif_weights_kernel, dim3(cdiv(B, R)), dim3(256), smem, (hipStream_t)stream, alphas, threshold, B,
                       T, cap, R, (f32x4*)weights, idx_out, mask_out, counts, max_count);
    STAR_LAUNCH_CHECK("star_cif_weights");
    return STAR_OK;
}

extern "C" int star_cif_aggregate_fwd(const void* s, int dtype, const float* weights, const int64_t* idx,
                                      const int32_t* counts, int B, int T, int cap, int d, void* out, void* stream) {
    using namespace star;
    STAR_REQUIRE(s && weights && idx && counts && out, "star_cif_aggregate_fwd: null pointer");
    const f32x4* w4 = (const f32x4*)weights;
    STAR_REQUIRE(B >= 1 && B <= 65535 && T >= 1 && cap >= 1 && d >= 1, "star_cif_aggregate_fwd: bad shape");
    STAR_REQUIRE(dtype == STAR_DT_BF16 || dtype == STAR_DT_F32, "star_cif_aggregate_fwd: dtype must be bf16 or f32");
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(cap, B);
    if (dtype == STAR_DT_BF16) {
        if (d % 8 == 0)
            hipLaunchKernelGGL((cif_aggregate_fwd_kernel<__bf16, 8>), grid, dim3(128), 0, st, (const __bf16*)s, w4,
                               idx, counts, T, cap, d, (__bf16*)out);
        else
            hipLaunchKernelGGL((cif_aggregate_fwd_kernel<__bf16, 1>), grid, dim3(128), 0, st, (const __bf16*)s, w4,
                               idx, counts, T, cap, d, (__bf16*)out);
    } else {
        if (d % 4 == 0)
            hipLaunchKernelGGL((cif_aggregate_fwd_kernel<float, 4>), grid, dim3(128), 0, st, (const float*)s, w4, idx,
                               counts, T, cap, d, (float*)out);
        else
            hipLaunchKernelGGL((cif_aggregate_fwd_kernel<float, 1>), grid, dim3(128), 0, st, (const float*)s, w4, idx,
                               counts, T, cap, d, (float*)out);
    }
    STAR_LAUNCH_CHECK("star_cif_aggregate_fwd");
    return STAR_OK;
}

extern "C" int star_cif_aggregate_bwd(const void* dout, const void* s, int dtype, const float* weights,
                                      const int32_t* counts, int B, int T, int cap, int d, void* ds,
                                      float* dalphas, float* workspace, void* stream) {
    using namespace star;
    STAR_REQUIRE(dout && s && weights && counts && ds && dalphas && workspace,
                 "star_cif_aggregate_bwd: null pointer");
    STAR_REQUIRE(B >= 1 && T >= 1 && cap >= 1 && d >= 1, "star_cif_aggregate_bwd: bad shape");
    STAR_REQUIRE(dtype == STAR_DT_BF16 || dtype == STAR_DT_F32, "star_cif_aggregate_bwd: dtype must be bf16 or f32");
    hipStream_t st = (hipStream_t)stream;
    float* dcur = workspace;
    float* drem = workspace + (size_t)B * T;
    const int64_t rows = (int64_t)B * T;
    const bool al16 = ((uintptr_t)dout % 16 == 0) && ((uintptr_t)s % 16 == 0) && ((uintptr_t)ds % 16 == 0);
#define STAR_CIF_BWD(TT, VV)                                                                                          \
    hipLaunchKernelGGL((cif_aggregate_bwd_kernel<TT, VV>), dim3(cdiv(rows, 4)), dim3(256), 0, st, (const TT*)dout,    \
                       (const TT*)s, (const f32x4*)weights, counts, B, T, cap, d, (TT*)ds, dcur, drem)
    if (dtype == STAR_DT_BF16) {
        if (d % 8 == 0 && al16) STAR_CIF_BWD(__bf16, 8);
        else STAR_CIF_BWD(__bf16, 1);
    } else {
        if (d % 4 == 0 && al16) STAR_CIF_BWD(float, 4);
        else STAR_CIF_BWD(float, 1);
    }
#undef STAR_CIF_BWD
    hipLaunchKernelGGL(cif_dalpha_kernel, dim3(B), dim3(256), 0, st, (const float*)dcur, (const float*)drem,
                       (const f32x4*)weights, T, dalphas);
    STAR_LAUNCH_CHECK("star_cif_aggregate_bwd");
    return STAR_OK;
}
