// CIF-style segmenter of the simultaneous path:
//   quantity loss      src/models/w2v_simul.py:419-424
//   resize             src/models/w2v_simul.py:504-535
//   fire scan          src/models/w2v_simul.py:376-403 (+ get_valid_mask_indices 196-220)
//   block mask         src/models/w2v_simul.py:382-402 (built lazily from the fire positions)
// All of it is HBM/latency-bound fp32 + integer work: B*T*4 bytes in, a few KB out.
#include "common.h"

namespace star {

// 1024 threads per utterance: these kernels are chains of block-wide phases over one row of T frames, and with 256
// threads every phase looped 6-12 times (cif_segment_kernel at T = 3000: 48 us, of which the serial scan is ~12).
constexpr int CIF_THREADS = 1024;
constexpr int CIF_WAVES = CIF_THREADS / 64;

// block-wide sum of a double, result broadcast to every thread; red must hold >= CIF_WAVES doubles.  Every kernel
// here sums with the same partition (thread i takes frames i, i + CIF_THREADS, ...; lanes, then waves in order).
__device__ __forceinline__ double block_sum_f64(double v, double* red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wid] = v;
    __syncthreads();
    double t = red[0];
#pragma unroll
    for (int w = 1; w < CIF_WAVES; ++w) t += red[w];
    return t;
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }

// ------------------------------------------------------------------------------------------ a5
__global__ __launch_bounds__(CIF_THREADS) void cif_quantity_fwd_kernel(
    const float* __restrict__ scores, const uint8_t* __restrict__ attn_mask, const int32_t* __restrict__ lengths,
    const int32_t* __restrict__ target_lengths, float threshold, int T, float* __restrict__ alphas,
    float* __restrict__ diff) {
    __shared__ double red[CIF_WAVES];
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* srow = scores + (size_t)b * T;
    const uint8_t* mrow = attn_mask ? attn_mask + (size_t)b * T : nullptr;
    const int n = lengths ? lengths[b] : 0;
    double acc = 0.0;
    for (int i = tid; i < T; i += CIF_THREADS) {
        bool valid = mrow ? (mrow[i] != 0) : (i < n);
        // accurate expf (not __expf): alphas feed bit-exact fire decisions downstream
        float a = valid ? 1.0f / (1.0f + expf(-srow[i])) : 0.0f;
        alphas[(size_t)b * T + i] = a;
        acc += (double)(a / threshold);
    }
    double tot = block_sum_f64(acc, red);
    if (tid == 0) diff[b] = (float)tot - (float)target_lengths[b];
}

__global__ __launch_bounds__(CIF_THREADS) void cif_quantity_loss_kernel(const float* __restrict__ diff, int B,
                                                                        float* __restrict__ loss) {
    __shared__ double red[CIF_WAVES];
    double acc = 0.0;
    for (int i = threadIdx.x; i < B; i += CIF_THREADS) acc += (double)diff[i] * (double)diff[i];
    double tot = block_sum_f64(acc, red);
    if (threadIdx.x == 0) loss[0] = (float)(tot / (double)B);
}

__global__ __launch_bounds__(CIF_THREADS) void cif_quantity_bwd_kernel(const float* __restrict__ alphas,
                                                                       const float* __restrict__ diff,
                                                                       const float* __restrict__ gloss,
                                                                       float threshold, int B, int T, int accumulate,
                                                                       float* __restrict__ dscores) {
    const int b = blockIdx.y;
    const int i = blockIdx.x * CIF_THREADS + threadIdx.x;
    if (i >= T) return;
    float a = alphas[(size_t)b * T + i];
    float g = gloss[0] * 2.0f * diff[b] / ((float)B * threshold) * a * (1.0f - a);
    size_t o = (size_t)b * T + i;
    dscores[o] = accumulate ? dscores[o] + g : g;
}

// ------------------------------------------------------------------------------------------ a6
// One workgroup per utterance; the row lives in LDS.  The reference's batch-global while-loop is
// row-separable (a fix-up only touches its own row), so every row runs its own <= max_rounds rounds.
__global__ __launch_bounds__(CIF_THREADS) void cif_resize_kernel(const float* __restrict__ alphas,
                                                                 const int32_t* __restrict__ target_lengths,
                                                                 float threshold, int max_rounds, int T,
                                                                 float* __restrict__ out,
                                                                 int32_t* __restrict__ fixups_out) {
    extern __shared__ float rs_smem[];
    float* row = rs_smem;                          // [T]
    uint32_t* snap = (uint32_t*)(row + T);         // [(T+31)/32] offender snapshot bits
    __shared__ double red[CIF_WAVES];
    __shared__ int flag;
    const int b = blockIdx.x, tid = threadIdx.x;
    const int nwords = (T + 31) / 32;
    double acc = 0.0;
    for (int i = tid; i < T; i += CIF_THREADS) {
        float a = alphas[(size_t)b * T + i];
        row[i] = a;
        acc += (double)a;
    }
    float den = (float)block_sum_f64(acc, red);
    float scale = __fdiv_rn((float)target_lengths[b], den);
    for (int i = tid; i < T; i += CIF_THREADS) row[i] = __fmul_rn(row[i], scale);
    __syncthreads();
    int nfix = 0;                                  // fix-ups applied to this row (uniform): the backward needs it
    for (int round = 0; round < max_rounds; ++round) {
        // snapshot of the offenders (value > thr), as torch.where does at the top of a round
        if (tid == 0) flag = 0;
        __syncthreads();
        for (int w = tid; w < nwords; w += CIF_THREADS) {
            uint32_t bits = 0;
            for (int k = 0; k < 32; ++k) {
                int i = w * 32 + k;
                if (i < T && row[i] > threshold) bits |= 1u << k;
            }
            snap[w] = bits;
            if (bits) flag = 1;
        }
        __syncthreads();
        if (!flag) break;
        for (int w = 0; w < nwords; ++w) {
            uint32_t bits = snap[w];  // uniform
            while (bits) {
                int y = w * 32 + __builtin_ctz(bits);
                bits &= bits - 1;
                if (row[y] >= threshold) {  // uniform: re-check the CURRENT value
                    ++nfix;
                    double s = 0.0, nz = 0.0;
                    for (int i = tid; i < T; i += CIF_THREADS) {
                        float a = row[i];
                        s += (double)a;
                        nz += (a != 0.0f) ? 1.0 : 0.0;
                    }
                    float sum = (float)block_sum_f64(s, red);
                    float cnt = (float)block_sum_f64(nz, red);
                    float mean = __fdiv_rn(__fmul_rn(0.5f, sum), cnt);
                    for (int i = tid; i < T; i += CIF_THREADS) {
                        float a = row[i];
                        row[i] = __fadd_rn(__fmul_rn(a, 0.5f), (a != 0.0f) ? mean : 0.0f);
                    }
                    __syncthreads();
                }
            }
        }
        __syncthreads();
    }
    for (int i = tid; i < T; i += CIF_THREADS) out[(size_t)b * T + i] = row[i];
    if (fixups_out && tid == 0) fixups_out[b] = nfix;
}

// Backward of resize (the gradient lm_cif_training sends through it, w2v_simul.py:558 / w2v_asr.py:679).
// Forward per row: r0 = a * c, c = L / S, S = sum(a); then k fix-ups r <- 0.5 r + (0.5 sum(r) / n) m with
// m = (r != 0), n = |m| (constants of the graph, w2v_simul.py:529-532).  m does not change between fix-ups (zeros
// stay zero; non-zero entries of a sigmoid output stay positive), and s = <m, g> is invariant under the transposed
// fix-up g <- 0.5 g + (0.5 <m, g> / n) 1, so k of them collapse to  g_k = 2^-k g + (s / n)(1 - 2^-k) 1.
// Then d a_i = c (g_k,i - <g_k, a> / S).  Sums in fp64 (fixed order), one workgroup per row.
__global__ __launch_bounds__(CIF_THREADS) void cif_resize_bwd_kernel(const float* __restrict__ alphas,
                                                                     const int32_t* __restrict__ target_lengths,
                                                                     const int32_t* __restrict__ fixups,
                                                                     const float* __restrict__ dout, int T,
                                                                     float* __restrict__ dalphas) {
    __shared__ double red[CIF_WAVES];
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* a = alphas + (size_t)b * T;
    const float* g = dout + (size_t)b * T;
    double S = 0.0, sg = 0.0, nz = 0.0, ga = 0.0, asum = 0.0;
    for (int i = tid; i < T; i += CIF_THREADS) {
        const double ai = a[i], gi = g[i];
        S += ai;
        if (a[i] != 0.0f) {
            sg += gi;
            nz += 1.0;
        }
        ga += gi * ai;
    }
    S = block_sum_f64(S, red);
    sg = block_sum_f64(sg, red);
    nz = block_sum_f64(nz, red);
    ga = block_sum_f64(ga, red);
    asum = S;
    const int k = fixups ? fixups[b] : 0;
    const double w = k > 0 ? ldexp(1.0, -min(k, 1000)) : 1.0;            // 2^-k
    const double add = (k > 0 && nz > 0.0) ? (sg / nz) * (1.0 - w) : 0.0;
    // <g_k, a> = w <g, a> + add * sum(a)
    const double dot = w * ga + add * asum;
    const double c = (double)target_lengths[b] / S;
    for (int i = tid; i < T; i += CIF_THREADS) {
        const double gk = w * (double)g[i] + add;
        dalphas[(size_t)b * T + i] = (float)(c * (gk - dot / S));
    }
}

// ------------------------------------------------------------------------------------------ a7
// The recurrence is inherently serial in t and fp32 addition is not associative, so a parallel prefix sum cannot
// reproduce the reference's fire positions: every utterance runs its own exact serial recurrence (one lane each).
// Everything around it is parallel: a workgroup owns a few utterances, brings their rows (alphas * mask) and the
// per-frame thresholds (1; -inf at the last valid frame = the forced fire) into LDS with all 256 threads, lane r of
// wave 0 runs cif_serial_chain() on row r (4 instructions per frame; the row becomes the pre-fire integrals), and all
// threads then turn integrals into fire positions by an order-preserving compaction.
// History at B = 32, T = 1500: 212 us (64 utterances per wave on LDS-transposed tiles), 88 us (one lane per row, but
// thresholds, bounds tests and fire-bit bookkeeping inside the serial loop: ~280 instructions per 16 frames).

// Rows per workgroup: two arrays of R * pitch floats in LDS.
__host__ __device__ static inline int scan_pitch(int T) { return ((T + 15) & ~15) + 4; }
// One utterance per workgroup unless the batch is huge: the parallel phases (row load, compaction) then cost one row
// per workgroup instead of R rows one after the other, and the serial chains still all run at the same time.
static inline int scan_rows_per_wg(int B, int T) {
    int r = 8;
    while (r > 1 && ((size_t)r * scan_pitch(T) * 8 > 128 * 1024 || (B + r - 1) / r < 2048)) r >>= 1;
    return r;
}

// state_cur / state_pos (both null for the whole-utterance scan): the streaming form of row f4.  The T frames are the
// chunk that follows state_pos[b] frames already consumed; lengths[] stays the utterance length in absolute frames,
// positions are written as absolute frame numbers, the integrator is carried in from and out to state_cur[b].
__global__ __launch_bounds__(256) void cif_fire_scan_kernel(
    const float* __restrict__ alphas, const uint8_t* __restrict__ attn_mask, const int32_t* __restrict__ lengths,
    const float* __restrict__ scores, int B, int T, int cap, int R, int64_t* __restrict__ idx_out,
    uint8_t* __restrict__ mask_out, int32_t* __restrict__ counts, int32_t* __restrict__ max_count,
    float* __restrict__ scores_gather, float* __restrict__ state_cur, int32_t* __restrict__ state_pos) {
    extern __shared__ __attribute__((aligned(16))) float fs_rows[];       // [R][pitch] alphas * mask -> integrals
    __shared__ int lens[8], wtot[4], slow[8];
    const int pitch = scan_pitch(T), T16 = (T + 15) & ~15;
    float* thr_rows = fs_rows + (size_t)R * pitch;                        // [R][pitch] per-frame thresholds
    float* carries = thr_rows + (size_t)R * pitch;                        // [R][pitch / 16] carry into every 16-frame block
    __shared__ int unit_row[8];
    const int tid = threadIdx.x;
    const int b0 = blockIdx.x * R;
    const int nrows = min(R, B - b0);
    if (tid < 8) {
        lens[tid] = 0;
        slow[tid] = 0;
    }
    __syncthreads();
    // phase 1 (all threads, coalesced): rows -> LDS, valid-frame counts (== attn_mask.sum(1) of the reference)
    for (int r = 0; r < nrows; ++r) {
        const size_t base = (size_t)(b0 + r) * T;
        const int pos0 = state_pos ? state_pos[b0 + r] : 0;
        // frames of this chunk that are valid: [0, n_r)
        const int n_r = attn_mask ? 0 : max(min((int64_t)lengths[b0 + r] - pos0, (int64_t)T), (int64_t)0);
        int c = 0;
        bool odd = false;                                  // a weight outside {0} u [FLT_MIN, 1] (NaN included)
        for (int t = tid; t < T16; t += 256) {
            float v = 0.0f;
            if (t < T) {
                const float a = alphas[base + t];
                const bool valid = attn_mask ? (attn_mask[base + t] != 0) : (t < n_r);
                v = valid ? a : a * 0.0f;                  // alphas * attn_mask (keeps NaN/inf of the product)
                c += valid ? 1 : 0;
                odd |= !(v == 0.0f || (v >= 1.17549435e-38f && v < 1.0f));
            }
            fs_rows[r * pitch + t] = v;
        }
        if (odd) slow[r] = 1;
        if (attn_mask) {
            c = wave_sum(c);
            if ((tid & 63) == 0) atomicAdd(&lens[r], c);
        } else if (tid == 0) {
            // unclamped: a length beyond the last frame means "every frame valid and NO forced last-frame fire" (the
            // threshold row below then has no -inf inside [0, T); get_valid_mask_indices and the streaming step rely
            // on it).  Relative to this chunk; clamped below so that the int cannot overflow.
            const int64_t rel = (int64_t)lengths[b0 + r] - pos0;
            lens[r] = (int)max(min(rel, (int64_t)T + 1), (int64_t)-1);
        }
    }
    __syncthreads();
    for (int r = 0; r < nrows; ++r) {
        const int n = lens[r];
        bool late = false;                                 // a non-zero weight after the forced fire (mask with holes)
        for (int t = tid; t < T16; t += 256) {
            // frames T..T16 are padding of the LDS row, not frames of the utterance: they never fire, so the carry
            // handed to the next chunk is the carry after frame T-1
            thr_rows[r * pitch + t] = t >= T ? __builtin_inff() : ((n == t + 1) ? -__builtin_inff() : 1.0f);
            late |= (t >= n && n >= 1 && fs_rows[r * pitch + t] != 0.0f);
        }
        if (late) slow[r] = 1;
    }
    __syncthreads();
    // phase 2: lane r of wave 0 runs the exact serial fp32 recurrence of row r.  On the common (two-instruction) path
    // the lane keeps only the carry into every 16-frame block and all threads rebuild the pre-fire integrals afterwards
    // (cif_serial_carries_unit / cif_expand_block_unit: 12.3 instead of 19.1 cycles per frame on the serial lane)
    const int cpitch = pitch >> 4;
    float c1 = 0.0f;
    bool unit = false;
    if (tid < nrows) {
        const float c0 = state_cur ? state_cur[b0 + tid] : 0.0f;
        unit = !slow[tid] && (c0 == 0.0f || (c0 >= 1.17549435e-38f && c0 < 1.0f));
        unit_row[tid] = unit ? 1 : 0;
        if (unit) c1 = cif_serial_carries_unit(fs_rows + tid * pitch, carries + tid * cpitch, T16 >> 4, c0);
        else c1 = cif_serial_chain<true>(fs_rows + tid * pitch, thr_rows + tid * pitch, 1.0f, T16 >> 4, c0);
    }
    __syncthreads();
    for (int r = 0; r < nrows; ++r) {
        if (!unit_row[r]) continue;
        for (int blk = tid; blk < (T16 >> 4); blk += 256) cif_expand_block_unit(fs_rows + r * pitch, carries + r * cpitch, blk);
    }
    __syncthreads();
    if (tid < nrows) {
        if (unit) {
            // the forced fire subtracts 1 whatever the integral was; every later weight of the chunk is zero
            const int n = lens[tid];
            if (n >= 1 && n <= T) c1 = __fsub_rn(fs_rows[tid * pitch + n - 1], 1.0f);
        }
        if (state_cur) state_cur[b0 + tid] = c1;
    }
    // phase 3 (all threads): fire_t = (I_t >= thr_t) for t < T; ascending positions by an exclusive scan over
    // contiguous chunks of frames; pad slots carry index -1 == frame T-1
    const int C = (T + 255) / 256;
    for (int r = 0; r < nrows; ++r) {
        const int b = b0 + r;
        const int pos0 = state_pos ? state_pos[b] : 0;
        const float* ir = fs_rows + r * pitch;
        const float* tr = thr_rows + r * pitch;
        const int lo = min(tid * C, T), hi = min(lo + C, T);
        int cnt = 0;
        for (int t = lo; t < hi; ++t) cnt += (ir[t] >= tr[t]) ? 1 : 0;
        int nf;
        int slot = block_excl_scan_256(cnt, wtot, nf);
        for (int t = lo; t < hi; ++t) {
            if (ir[t] >= tr[t]) {
                if (slot < cap) {
                    idx_out[(size_t)b * cap + slot] = (int64_t)pos0 + t;
                    if (mask_out) mask_out[(size_t)b * cap + slot] = 1;
                    if (scores_gather) scores_gather[(size_t)b * cap + slot] = scores[(size_t)b * T + t];
                }
                ++slot;
            }
        }
        if (tid == 0) {
            if (counts) counts[b] = nf;
            if (max_count) atomicMax(max_count, nf);
        }
        const float last_score = scores_gather ? scores[(size_t)b * T + (T - 1)] : 0.0f;
        for (int j = nf + tid; j < cap; j += 256) {
            idx_out[(size_t)b * cap + j] = -1;
            if (mask_out) mask_out[(size_t)b * cap + j] = 0;
            if (scores_gather) scores_gather[(size_t)b * cap + j] = last_score;
        }
    }
    __syncthreads();                                       // every thread has read state_pos[b] before it moves
    if (state_pos && tid < nrows) state_pos[b0 + tid] += T;
}

// ------------------------------------------------------------------------------------------ a5 + a6 + a7 + a3 fused
// The simultaneous path's segmenter in ONE launch (w2v_simul.py:419-431): alphas = sigmoid(scores) * mask and the
// quantity-loss difference (a5), resize (a6), the fire scan with its score gather (a7) and the nugget type table (a3),
// one workgroup per utterance with the row resident in LDS from the first step to the last.  Every stage is the code
// of the stand-alone kernel above (same thread partition of every sum, same recurrence), so all outputs are
// bit-identical to cif_quantity_fwd -> cif_resize -> cif_fire_scan -> gather -> type table; it saves five launches of
// 4-9 us each on the critical path of the step.  Prefix masks only (lengths).
__global__ __launch_bounds__(CIF_THREADS) void cif_segment_kernel(
    const float* __restrict__ scores, const int32_t* __restrict__ lengths, const int32_t* __restrict__ target_lengths,
    float threshold, int max_rounds, int T, int cap, float* __restrict__ alphas, float* __restrict__ diff,
    float* __restrict__ resized, int32_t* __restrict__ fixups_out, int64_t* __restrict__ idx_out,
    uint8_t* __restrict__ mask_out, int32_t* __restrict__ counts, int32_t* __restrict__ max_count,
    float* __restrict__ scores_gather, int64_t* __restrict__ table, int64_t* __restrict__ positions) {
    extern __shared__ __attribute__((aligned(16))) float sg_smem[];
    const int pitch = scan_pitch(T), T16 = (T + 15) & ~15;
    float* row = sg_smem;                          // [pitch] alphas -> resized -> pre-fire integrals
    float* thr_row = sg_smem + pitch;              // [pitch] per-frame thresholds of the scan
    uint32_t* snap = (uint32_t*)(sg_smem + 2 * pitch);   // resize: offender snapshot bits
    float* carry = sg_smem + 2 * pitch + ((T + 31) / 32);     // [T16 / 16] fire scan: the carry into every 16-frame block
    __shared__ double red[CIF_WAVES];
    __shared__ int flag, wtot[CIF_WAVES], slow;
    const int b = blockIdx.x, tid = threadIdx.x;
    const int n = lengths[b];                      // unclamped (a length > T: every frame valid, no forced fire)
    const float* srow = scores + (size_t)b * T;
    const int nwords = (T + 31) / 32;
    if (tid == 0) slow = 0;

    // ---- a5: alphas and the quantity difference (cif_quantity_fwd_kernel)
    double acc = 0.0, acc_r = 0.0;
    for (int i = tid; i < T; i += CIF_THREADS) {
        const float a = (i < n) ? 1.0f / (1.0f + expf(-srow[i])) : 0.0f;
        alphas[(size_t)b * T + i] = a;
        row[i] = a;
        acc += (double)(a / threshold);
        acc_r += (double)a;
    }
    for (int i = T + tid; i < T16; i += CIF_THREADS) row[i] = 0.0f;
    const double tot = block_sum_f64(acc, red);
    if (tid == 0) diff[b] = (float)tot - (float)target_lengths[b];

    // ---- a6: resize (cif_resize_kernel)
    const float den = (float)block_sum_f64(acc_r, red);
    const float scale = __fdiv_rn((float)target_lengths[b], den);
    for (int i = tid; i < T; i += CIF_THREADS) row[i] = __fmul_rn(row[i], scale);
    __syncthreads();
    int nfix = 0;
    for (int round = 0; round < max_rounds; ++round) {
        if (tid == 0) flag = 0;
        __syncthreads();
        // one ballot per 64 frames (the same bits as cif_resize_kernel's word loop, without its 32-way bank conflict)
        for (int i0 = 0; i0 < nwords * 32; i0 += CIF_THREADS) {
            const int i = i0 + tid;
            const unsigned long long bal = __ballot(i < T && row[i] > threshold);
            if ((tid & 63) == 0) {
                if ((i >> 5) < nwords) snap[i >> 5] = (uint32_t)bal;
                if ((i >> 5) + 1 < nwords) snap[(i >> 5) + 1] = (uint32_t)(bal >> 32);
                if (bal) flag = 1;
            }
        }
        __syncthreads();
        if (!flag) break;
        for (int w = 0; w < nwords; ++w) {
            uint32_t bits = snap[w];
            while (bits) {
                const int y = w * 32 + __builtin_ctz(bits);
                bits &= bits - 1;
                if (row[y] >= threshold) {
                    ++nfix;
                    double s = 0.0, nz = 0.0;
                    for (int i = tid; i < T; i += CIF_THREADS) {
                        const float a = row[i];
                        s += (double)a;
                        nz += (a != 0.0f) ? 1.0 : 0.0;
                    }
                    const float sum = (float)block_sum_f64(s, red);
                    const float cnt = (float)block_sum_f64(nz, red);
                    const float mean = __fdiv_rn(__fmul_rn(0.5f, sum), cnt);
                    for (int i = tid; i < T; i += CIF_THREADS) {
                        const float a = row[i];
                        row[i] = __fadd_rn(__fmul_rn(a, 0.5f), (a != 0.0f) ? mean : 0.0f);
                    }
                    __syncthreads();
                }
            }
        }
        __syncthreads();
    }
    if (fixups_out && tid == 0) fixups_out[b] = nfix;

    // ---- a7: fire scan on the resized row (cif_fire_scan_kernel, one row per workgroup)
    bool odd = false;
    for (int t = tid; t < T16; t += CIF_THREADS) {
        float v = 0.0f;
        if (t < T) {
            const float a = row[t];
            resized[(size_t)b * T + t] = a;
            v = (t < n) ? a : a * 0.0f;                    // resized * mask (keeps NaN / inf of the product)
            odd |= !(v == 0.0f || (v >= 1.17549435e-38f && v < 1.0f));
        }
        row[t] = v;
        thr_row[t] = t >= T ? __builtin_inff() : ((n == t + 1) ? -__builtin_inff() : 1.0f);
    }
    if (odd) slow = 1;
    __syncthreads();
    // the serial lane leaves only block carries on the common path; all threads then rebuild the pre-fire integrals
    // (cif_serial_carries_unit: 12.3 instead of 19.1 cycles per frame on the one lane everybody waits for)
    if (tid == 0) {
        if (!slow) cif_serial_carries_unit(row, carry, T16 >> 4, 0.0f);
        else cif_serial_chain<true>(row, thr_row, 1.0f, T16 >> 4, 0.0f);
    }
    __syncthreads();
    if (!slow) {
        for (int blk = tid; blk < (T16 >> 4); blk += CIF_THREADS) cif_expand_block_unit(row, carry, blk);
        __syncthreads();
    }
    const int C = (T + CIF_THREADS - 1) / CIF_THREADS;
    const int lo = min(tid * C, T), hi = min(lo + C, T);
    int cnt = 0;
    for (int t = lo; t < hi; ++t) cnt += (row[t] >= thr_row[t]) ? 1 : 0;
    int nf;
    int slot = block_excl_scan<CIF_WAVES>(cnt, wtot, nf);
    for (int t = lo; t < hi; ++t) {
        const bool fired = row[t] >= thr_row[t];
        if (fired) {
            if (slot < cap) {
                idx_out[(size_t)b * cap + slot] = t;
                mask_out[(size_t)b * cap + slot] = 1;
                scores_gather[(size_t)b * cap + slot] = srow[t];
                if (positions) positions[(size_t)b * cap + slot] = t;
            }
            ++slot;
        }
        // ---- a3: the type table marks the frames whose fire has a slot (star_nugget_type_table on idx / mask)
        if (table) table[(size_t)b * T + t] = (fired && slot <= cap) ? 1 : 0;
    }
    if (tid == 0) {
        counts[b] = nf;
        if (max_count) atomicMax(max_count, nf);
    }
    const float last_score = srow[T - 1];
    for (int j = nf + tid; j < cap; j += CIF_THREADS) {
        idx_out[(size_t)b * cap + j] = -1;
        mask_out[(size_t)b * cap + j] = 0;
        scores_gather[(size_t)b * cap + j] = last_score;
        if (positions) positions[(size_t)b * cap + j] = -1;
    }
}

// ------------------------------------------------------------------------------------------ f4: one launch per chunk
// The streaming segmenter's whole per-chunk tail in ONE kernel, one workgroup per
// utterance: alphas = sigmoid(score) * mask (w2v_simul.py:420-421, the expression of cif_quantity_fwd_kernel, so the
// weights are the same bits), the fire recurrence carried across chunks (cif_fire_scan_kernel's streaming form), and
// the append of the fired semantic rows, their scorer logits (w2v_simul.py:427: the attention bias) and their absolute
// positions (w2v_simul.py:446-447) to the utterance's memory at counts[b].  Tc == 0 runs the end-of-utterance flush
// only.  Row copies are 16 bytes per lane (the host checks d * esize % 16 == 0 and the alignment).
struct StreamMem {
    unsigned char* mem;        // [B][kmax][row_bytes]
    float* mem_scores;         // [B][kmax]
    int64_t* mem_pos;          // [B][kmax]
    int32_t* counts;           // [B] in/out, never above kmax
    int32_t* dropped;          // [B] in/out: fires that found no slot
    unsigned char* last_row;   // [B][row_bytes]: the newest frame's semantic row (what a late flush appends)
    float* last_score;         // [B]
    int32_t* state_last;       // [B] in/out: absolute position of the newest fire, -1 = none yet
    int kmax;
    int row_bytes;
};

__device__ __forceinline__ void stream_copy_row(unsigned char* dst, const unsigned char* src, int row_bytes, int lane) {
    for (int o = lane * 16; o < row_bytes; o += 64 * 16) *(u32x4*)(dst + o) = *(const u32x4*)(src + o);
}

__global__ __launch_bounds__(256) void cif_stream_push_kernel(
    const float* __restrict__ scores, const unsigned char* __restrict__ s_chunk, const int32_t* __restrict__ lengths,
    int Tc, float* __restrict__ state_cur, int32_t* __restrict__ state_pos, StreamMem m, int cap,
    int64_t* __restrict__ idx_out, int32_t* __restrict__ cnt_out) {
    extern __shared__ __attribute__((aligned(16))) float sp_smem[];
    const int pitch = scan_pitch(Tc), T16 = (Tc + 15) & ~15;
    float* row = sp_smem;                       // [pitch] alphas -> pre-fire integrals
    float* thr_row = sp_smem + pitch;           // [pitch]
    int* fire_list = (int*)(sp_smem + 2 * pitch);   // [Tc] chunk-relative fire positions, ascending
    __shared__ int wtot[4], slow;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int pos0 = state_pos[b];
    const int len = lengths[b];
    int c0 = m.counts[b];
    int last = m.state_last[b];
    float cur0 = state_cur[b];
    int lost = 0;
    if (tid == 0) slow = 0;
    __syncthreads();
    // ---- the end of the utterance became known only AFTER its last frame was pushed (len == frames consumed): the
    // forced fire of that frame (w2v_simul.py:388, src_len == i + 1) is emitted now from the saved row, unless the
    // frame fired by itself
    if (len == pos0 && pos0 > 0 && last != len - 1) {
        if (c0 < m.kmax) {
            if (wid == 0)
                stream_copy_row(m.mem + ((size_t)b * m.kmax + c0) * m.row_bytes, m.last_row + (size_t)b * m.row_bytes,
                                m.row_bytes, lane);
            if (tid == 0) {
                m.mem_scores[(size_t)b * m.kmax + c0] = m.last_score[b];
                m.mem_pos[(size_t)b * m.kmax + c0] = (int64_t)len - 1;
            }
            ++c0;
        } else {
            ++lost;
        }
        last = len - 1;
        cur0 = __fsub_rn(cur0, 1.0f);           // the reference subtracts 1 at every fire, forced or not
    }
    int nf = 0;
    if (Tc > 0) {
        const float* srow = scores + (size_t)b * Tc;
        // frames of this chunk inside the utterance: [0, n)
        const int n = (int)max(min((int64_t)len - pos0, (int64_t)Tc), (int64_t)0);
        const int64_t rel = (int64_t)len - pos0;                       // forced fire at chunk frame rel - 1
        bool odd = false;
        for (int t = tid; t < T16; t += 256) {
            float v = 0.0f;
            if (t < n) v = 1.0f / (1.0f + expf(-srow[t]));             // cif_quantity_fwd_kernel's expression
            odd |= !(v == 0.0f || (v >= 1.17549435e-38f && v < 1.0f));
            row[t] = v;
            thr_row[t] = t >= Tc ? __builtin_inff() : ((rel == (int64_t)t + 1) ? -__builtin_inff() : 1.0f);
        }
        if (odd) slow = 1;
        __syncthreads();
        if (tid == 0) {
            const bool unit = !slow && (cur0 == 0.0f || (cur0 >= 1.17549435e-38f && cur0 < 1.0f));
            float c1;
            if (unit) {
                c1 = cif_serial_chain_unit(row, T16 >> 4, cur0);
                if (rel >= 1 && rel <= Tc) c1 = __fsub_rn(row[rel - 1], 1.0f);
            } else {
                c1 = cif_serial_chain<true>(row, thr_row, 1.0f, T16 >> 4, cur0);
            }
            state_cur[b] = c1;
        }
        __syncthreads();
        const int C = (Tc + 255) / 256;
        const int lo = min(tid * C, Tc), hi = min(lo + C, Tc);
        int cnt = 0;
        for (int t = lo; t < hi; ++t) cnt += (row[t] >= thr_row[t]) ? 1 : 0;
        int slot = block_excl_scan_256(cnt, wtot, nf);
        for (int t = lo; t < hi; ++t)
            if (row[t] >= thr_row[t]) fire_list[slot++] = t;
        __syncthreads();
        // ---- append: one wave per fired row
        for (int j = wid; j < nf; j += 4) {
            const int t = fire_list[j];
            const int dest = c0 + j;
            if (dest < m.kmax) {
                stream_copy_row(m.mem + ((size_t)b * m.kmax + dest) * m.row_bytes,
                                s_chunk + ((size_t)b * Tc + t) * m.row_bytes, m.row_bytes, lane);
                if (lane == 0) {
                    m.mem_scores[(size_t)b * m.kmax + dest] = srow[t];
                    m.mem_pos[(size_t)b * m.kmax + dest] = (int64_t)pos0 + t;
                }
            }
        }
        if (idx_out) {
            for (int j = tid; j < cap; j += 256) idx_out[(size_t)b * cap + j] = j < nf ? (int64_t)pos0 + fire_list[j] : -1;
        }
        // the newest frame's row and logit, for a flush that comes with the next call
        if (wid == 3)
            stream_copy_row(m.last_row + (size_t)b * m.row_bytes, s_chunk + ((size_t)b * Tc + (Tc - 1)) * m.row_bytes,
                            m.row_bytes, lane);
        if (nf > 0) last = pos0 + fire_list[nf - 1];
        if (tid == 0) m.last_score[b] = srow[Tc - 1];
    } else if (tid == 0) {
        state_cur[b] = cur0;
    }
    if (cnt_out && tid == 0) cnt_out[b] = nf;
    if (tid == 0) {
        const int total = c0 + nf;
        m.counts[b] = min(total, m.kmax);
        m.dropped[b] += lost + max(total - m.kmax, 0);
        m.state_last[b] = last;
        state_pos[b] = pos0 + Tc;
    }
}

// block_attn_mask rows: out[b,i,j] = (prev_fire(i) < j <= i) && valid(i)
__global__ __launch_bounds__(256) void cif_block_mask_kernel(const int64_t* __restrict__ idx,
                                                             const int32_t* __restrict__ counts,
                                                             const uint8_t* __restrict__ attn_mask,
                                                             const int32_t* __restrict__ lengths, int T, int cap,
                                                             float* __restrict__ out) {
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= T) return;
    const int cnt = min(counts[b], cap);
    const int64_t* irow = idx + (size_t)b * cap;
    // prev = largest fire position < i  (binary search over the ascending positions)
    int lo = 0, hi = cnt;
    while (lo < hi) {
        int mid = (lo + hi) >> 1;
        if (irow[mid] < i) lo = mid + 1; else hi = mid;
    }
    const int prev = lo > 0 ? (int)irow[lo - 1] : -1;
    const bool valid = attn_mask ? (attn_mask[(size_t)b * T + i] != 0) : (i < lengths[b]);
    float* orow = out + ((size_t)b * T + i) * T;
    for (int j = lane; j < T; j += 64) orow[j] = (valid && j > prev && j <= i) ? 1.0f : 0.0f;
}

}  // namespace star

extern "C" int star_cif_quantity_fwd(const float* scores, const uint8_t* attn_mask, const int32_t* lengths,
                                     const int32_t* target_lengths, float threshold, int B, int T, float* alphas,
                                     float* diff, float* loss, void* stream) {
    using namespace star;
    STAR_REQUIRE(scores && target_lengths && alphas && diff && loss, "star_cif_quantity_fwd: null pointer");
    STAR_REQUIRE((attn_mask != nullptr) != (lengths != nullptr), "star_cif_quantity_fwd: give exactly one of attn_mask / lengths");
    STAR_REQUIRE(B >= 1 && T >= 1 && threshold > 0.f, "star_cif_quantity_fwd: bad shape B=%d T=%d", B, T);
    hipLaunchKernelGGL(cif_quantity_fwd_kernel, dim3(B), dim3(CIF_THREADS), 0, (hipStream_t)stream, scores, attn_mask,
                       lengths, target_lengths, threshold, T, alphas, diff);
    hipLaunchKernelGGL(cif_quantity_loss_kernel, dim3(1), dim3(CIF_THREADS), 0, (hipStream_t)stream, diff, B, loss);
    STAR_LAUNCH_CHECK("star_cif_quantity_fwd");
    return STAR_OK;
}

extern "C" int star_cif_quantity_bwd(const float* alphas, const float* diff, const float* gloss, float threshold,
                                     int B, int T, int accumulate, float* dscores, void* stream) {
    using namespace star;
    STAR_REQUIRE(alphas && diff && gloss && dscores, "star_cif_quantity_bwd: null pointer");
    STAR_REQUIRE(B >= 1 && B <= 65535 && T >= 1, "star_cif_quantity_bwd: bad shape B=%d T=%d", B, T);
    hipLaunchKernelGGL(cif_quantity_bwd_kernel, dim3(cdiv(T, CIF_THREADS), B), dim3(CIF_THREADS), 0,
                       (hipStream_t)stream, alphas, diff, gloss, threshold, B, T, accumulate, dscores);
    STAR_LAUNCH_CHECK("star_cif_quantity_bwd");
    return STAR_OK;
}

extern "C" int star_cif_resize(const float* alphas, const int32_t* target_lengths, float threshold, int max_rounds,
                               int B, int T, float* out, int32_t* fixups_out, void* stream) {
    using namespace star;
    STAR_REQUIRE(alphas && target_lengths && out, "star_cif_resize: null pointer");
    STAR_REQUIRE(B >= 1 && T >= 1 && T <= 16384, "star_cif_resize: bad shape B=%d T=%d (T <= 16384)", B, T);
    size_t smem = (size_t)T * 4 + (size_t)((T + 31) / 32) * 4;
    if (smem > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)cif_resize_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)smem);
        if (e != hipSuccess) {
            set_error("star_cif_resize: cannot reserve %zu bytes of LDS: %s", smem, hipGetErrorString(e));
            return STAR_ERR_LAUNCH;
        }
    }
    hipLaunchKernelGGL(cif_resize_kernel, dim3(B), dim3(CIF_THREADS), smem, (hipStream_t)stream, alphas,
                       target_lengths, threshold, max_rounds, T, out, fixups_out);
    STAR_LAUNCH_CHECK("star_cif_resize");
    return STAR_OK;
}

extern "C" int star_cif_resize_bwd(const float* alphas, const int32_t* target_lengths, const int32_t* fixups,
                                   const float* dout, int B, int T, float* dalphas, void* stream) {
    using namespace star;
    STAR_REQUIRE(alphas && target_lengths && dout && dalphas, "star_cif_resize_bwd: null pointer");
    STAR_REQUIRE(B >= 1 && T >= 1, "star_cif_resize_bwd: bad shape B=%d T=%d", B, T);
    hipLaunchKernelGGL(cif_resize_bwd_kernel, dim3(B), dim3(CIF_THREADS), 0, (hipStream_t)stream, alphas,
                       target_lengths, fixups, dout, T, dalphas);
    STAR_LAUNCH_CHECK("star_cif_resize_bwd");
    return STAR_OK;
}

extern "C" int star_cif_fire_scan(const float* alphas, const uint8_t* attn_mask, const int32_t* lengths,
                                  const float* scores, int B, int T, int cap, int64_t* idx_out, uint8_t* mask_out,
                                  int32_t* counts, int32_t* max_count, float* scores_gather, void* stream) {
    using namespace star;
    STAR_REQUIRE(alphas && idx_out, "star_cif_fire_scan: null pointer");
    STAR_REQUIRE((attn_mask != nullptr) != (lengths != nullptr), "star_cif_fire_scan: give exactly one of attn_mask / lengths");
    STAR_REQUIRE(B >= 1 && T >= 1 && cap >= 1, "star_cif_fire_scan: bad shape B=%d T=%d cap=%d", B, T, cap);
    STAR_REQUIRE(!scores_gather || scores, "star_cif_fire_scan: scores_gather requested without scores");
    STAR_REQUIRE(T <= 16384, "star_cif_fire_scan: T=%d exceeds the 16384-frame LDS row limit", T);
    const int R = scan_rows_per_wg(B, T);
    const size_t smem = (size_t)R * scan_pitch(T) * 8 + (size_t)R * (scan_pitch(T) >> 4) * 4;
    if (smem > 48 * 1024) {
        static DevOnce once;
        if (!reserve_lds_once(once, (const void*)cif_fire_scan_kernel, 152 * 1024)) {
            set_error("star_cif_fire_scan: cannot reserve LDS");
            return STAR_ERR_LAUNCH;
        }
    }
    if (max_count) {
        hipError_t e = hipMemsetAsync(max_count, 0, sizeof(int32_t), (hipStream_t)stream);
        if (e != hipSuccess) {
            set_error("star_cif_fire_scan: memset failed: %s", hipGetErrorString(e));
            return STAR_ERR_LAUNCH;
        }
    }
    hipLaunchKernelGGL(cif_fire_scan_kernel, dim3(cdiv(B, R)), dim3(256), smem, (hipStream_t)stream, alphas, attn_mask,
                       lengths, scores, B, T, cap, R, idx_out, mask_out, counts, max_count, scores_gather,
                       (float*)nullptr, (int32_t*)nullptr);
    STAR_LAUNCH_CHECK("star_cif_fire_scan");
    return STAR_OK;
}

extern "C" int star_cif_fire_step(const float* alpha_chunk, const int32_t* lengths, const float* score_chunk, int B,
                                  int Tc, float* state_cur, int32_t* state_pos, int cap, int64_t* idx_out,
                                  int32_t* counts, float* scores_gather, void* stream) {
    using namespace star;
    STAR_REQUIRE(alpha_chunk && lengths && state_cur && state_pos && idx_out && counts, "star_cif_fire_step: null pointer");
    STAR_REQUIRE(B >= 1 && Tc >= 1 && cap >= 1, "star_cif_fire_step: bad shape B=%d Tc=%d cap=%d", B, Tc, cap);
    STAR_REQUIRE(!scores_gather || score_chunk, "star_cif_fire_step: scores_gather requested without score_chunk");
    STAR_REQUIRE(Tc <= 16384, "star_cif_fire_step: Tc=%d exceeds the 16384-frame LDS row limit", Tc);
    const int R = scan_rows_per_wg(B, Tc);
    const size_t smem = (size_t)R * scan_pitch(Tc) * 8 + (size_t)R * (scan_pitch(Tc) >> 4) * 4;
    if (smem > 48 * 1024) {
        static DevOnce once;
        if (!reserve_lds_once(once, (const void*)cif_fire_scan_kernel, 152 * 1024)) {
            set_error("star_cif_fire_step: cannot reserve LDS");
            return STAR_ERR_LAUNCH;
        }
    }
    hipLaunchKernelGGL(cif_fire_scan_kernel, dim3(cdiv(B, R)), dim3(256), smem, (hipStream_t)stream, alpha_chunk,
                       (const uint8_t*)nullptr, lengths, score_chunk, B, Tc, cap, R, idx_out, (uint8_t*)nullptr, counts,
                       (int32_t*)nullptr, scores_gather, state_cur, state_pos);
    STAR_LAUNCH_CHECK("star_cif_fire_step");
    return STAR_OK;
}

extern "C" int star_cif_segment(const float* scores, const int32_t* lengths, const int32_t* target_lengths,
                                float threshold, int max_rounds, int B, int T, int cap, float* alphas, float* diff,
                                float* loss, float* resized, int32_t* fixups_out, int64_t* idx_out, uint8_t* mask_out,
                                int32_t* counts, int32_t* max_count, float* scores_gather, int64_t* type_table,
                                int64_t* positions, void* stream) {
    using namespace star;
    STAR_REQUIRE(scores && lengths && target_lengths && alphas && diff && loss && resized && idx_out && mask_out && counts &&
                     scores_gather,
                 "star_cif_segment: null pointer");
    STAR_REQUIRE(B >= 1 && T >= 1 && cap >= 1 && threshold > 0.f, "star_cif_segment: bad shape B=%d T=%d cap=%d", B, T, cap);
    STAR_REQUIRE(T <= 16384, "star_cif_segment: T=%d exceeds the 16384-frame LDS row limit", T);
    const size_t smem = (size_t)2 * scan_pitch(T) * 4 + (size_t)((T + 31) / 32) * 4 + (size_t)((T + 15) / 16) * 4;
    if (smem > 48 * 1024) {
        static DevOnce once;
        if (!reserve_lds_once(once, (const void*)cif_segment_kernel, 152 * 1024)) {
            set_error("star_cif_segment: cannot reserve LDS");
            return STAR_ERR_LAUNCH;
        }
    }
    if (max_count) {
        hipError_t e = hipMemsetAsync(max_count, 0, sizeof(int32_t), (hipStream_t)stream);
        if (e != hipSuccess) {
            set_error("star_cif_segment: memset failed: %s", hipGetErrorString(e));
            return STAR_ERR_LAUNCH;
        }
    }
    hipLaunchKernelGGL(cif_segment_kernel, dim3(B), dim3(CIF_THREADS), smem, (hipStream_t)stream, scores, lengths, target_lengths,
                       threshold, max_rounds, T, cap, alphas, diff, resized, fixups_out, idx_out, mask_out, counts, max_count,
                       scores_gather, type_table, positions);
    hipLaunchKernelGGL(cif_quantity_loss_kernel, dim3(1), dim3(CIF_THREADS), 0, (hipStream_t)stream, diff, B, loss);
    STAR_LAUNCH_CHECK("star_cif_segment");
    return STAR_OK;
}

extern "C" int star_cif_block_mask(const int64_t* idx, const int32_t* counts, const uint8_t* attn_mask,
                                   const int32_t* lengths, int B, int T, int cap, float* out, void* stream) {
    using namespace star;
    STAR_REQUIRE(idx && counts && out, "star_cif_block_mask: null pointer");
    STAR_REQUIRE((attn_mask != nullptr) != (lengths != nullptr), "star_cif_block_mask: give exactly one of attn_mask / lengths");
    STAR_REQUIRE(B >= 1 && B <= 65535 && T >= 1 && cap >= 1, "star_cif_block_mask: bad shape B=%d T=%d cap=%d", B, T, cap);
    hipLaunchKernelGGL(cif_block_mask_kernel, dim3(cdiv(T, 4), B), dim3(256), 0, (hipStream_t)stream, idx, counts,
                       attn_mask, lengths, T, cap, out);
    STAR_LAUNCH_CHECK("star_cif_block_mask");
    return STAR_OK;
}

extern "C" int star_cif_stream_push(const float* score_chunk, const void* s_chunk, int dtype, const int32_t* lengths,
                                    int B, int Tc, int d, float* state_cur, int32_t* state_pos, int32_t* state_last,
                                    int kmax, void* mem, float* mem_scores, int64_t* mem_pos, int32_t* counts,
                                    int32_t* dropped, void* last_row, float* last_score, int cap, int64_t* idx_out,
                                    int32_t* cnt_out, void* stream) {
    using namespace star;
    STAR_REQUIRE(lengths && state_cur && state_pos && state_last && mem && mem_scores && mem_pos && counts && dropped &&
                     last_row && last_score,
                 "star_cif_stream_push: null pointer");
    STAR_REQUIRE(Tc == 0 || (score_chunk && s_chunk), "star_cif_stream_push: a chunk of %d frames without data", Tc);
    STAR_REQUIRE(dtype == STAR_DT_BF16 || dtype == STAR_DT_F32, "star_cif_stream_push: bad dtype %d", dtype);
    STAR_REQUIRE(B >= 1 && Tc >= 0 && Tc <= 8192 && d >= 1 && kmax >= 1, "star_cif_stream_push: bad shape B=%d Tc=%d d=%d kmax=%d",
                 B, Tc, d, kmax);
    STAR_REQUIRE(!idx_out || cap >= 1, "star_cif_stream_push: idx_out without cap");
    const int row_bytes = d * (dtype == STAR_DT_BF16 ? 2 : 4);
    STAR_REQUIRE(row_bytes % 16 == 0 && (((uintptr_t)s_chunk | (uintptr_t)mem | (uintptr_t)last_row) & 15) == 0,
                 "star_cif_stream_push: rows must be multiples of 16 bytes at 16-byte aligned addresses (d=%d)", d);
    const size_t smem = (size_t)2 * scan_pitch(Tc) * 4 + (size_t)(Tc > 0 ? Tc : 1) * 4;
    if (smem > 48 * 1024) {
        static DevOnce once;
        if (!reserve_lds_once(once, (const void*)cif_stream_push_kernel, 112 * 1024)) {
            set_error("star_cif_stream_push: cannot reserve LDS");
            return STAR_ERR_LAUNCH;
        }
    }
    StreamMem m = {(unsigned char*)mem, mem_scores, mem_pos, counts, dropped, (unsigned char*)last_row, last_score,
                   state_last, kmax, row_bytes};
    hipLaunchKernelGGL(cif_stream_push_kernel, dim3(B), dim3(256), smem, (hipStream_t)stream, score_chunk,
                       (const unsigned char*)s_chunk, lengths, Tc, state_cur, state_pos, m, cap, idx_out, cnt_out);
    STAR_LAUNCH_CHECK("star_cif_stream_push");
    return STAR_OK;
}
