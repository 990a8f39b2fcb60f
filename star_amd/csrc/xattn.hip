// Score-biased decoder cross-attention core, forward + backward.
//
// Reference: MyGPT2Attention._attn, cross-attention branch (src/models/my_transformer.py:535-603):
//     A = Q K^T / sqrt(hd)  [+ ILK mask: key j visible to query i iff j <= i+2]
//       + encoder pad mask + scorer_logits[b, j]              (582-589: the "nugget logits")
//     P = softmax(A),  O = P V
// and its autograd, whose d(scorer_logits)[b,j] = sum_{h,i} dS[b,h,i,j] is the cross-attention
// feedback into the scorer.  The reference materialises three [B,h,L,K] tensors per layer and builds
// the ILK mask on the CPU; here everything is flash-style on bf16 MFMA (32x32x16), fp32 softmax:
//
//  forward   one workgroup = 4 waves x 32 queries; K/V tiles of 64 keys staged in LDS.
//            S^T = K Q^T (query on the MFMA lane -> row max / row sum are in-lane), the P^T
//            accumulator is re-used directly as the B operand of O^T = V^T P^T, V^T fragments come
//            from ds_read_b64_tr_b16 on the row-major V tile.  Saves LSE for the backward.
//  backward  delta = rowsum(dO*O);  dQ kernel (same walk as forward, recomputes P from LSE);
//            dK/dV kernel: one wave = 32 keys with dK^T, dV^T in accumulators, key on the lane so the
//            S and dP accumulators are the B operands of the dV^T / dK^T products; the per-key
//            column sum of dS is the bias gradient, reduced over heads in a fixed order.
//  No atomics anywhere: results are bitwise reproducible.
// Masked logits are -inf (P and dS exactly 0), the running max starts at a finite sentinel.
// Algorithmic FLOP per (b, layer): forward 4*L*k_b*d, backward 8*L*k_b*d (+ 4*L*k_b*d recompute).
#include "common.h"

namespace star {

constexpr int XA_THREADS = 256;
constexpr int KT = 64;          // keys per LDS tile (forward, dQ)
constexpr float NEG_SENTINEL = -1.0e30f;
#ifndef XA_ABLATE
#define XA_ABLATE 0
#endif

__device__ __forceinline__ int crow(int reg, int half) { return (reg & 3) + 8 * (reg >> 2) + 4 * half; }

// LDS pitch (bytes) of a [rows][HD] bf16 tile read by ROWS with ds_read_b128: +16 B per row
template <int HD> struct Pitch {
    static constexpr int ROW = HD * 2 + 16;
    // read TRANSPOSED with ds_read_b64_tr_b16 (4 consecutive rows x 64 B per half-wave): pitch = 64 or 192 mod 256
    static constexpr int TR = (HD == 128) ? 320 : 192;
};

__device__ __forceinline__ bf16x8 lds_row8(const uint8_t* p) { return *(const bf16x8*)p; }

__device__ __forceinline__ bf16x8 lds_tr8(const uint8_t* p0, const uint8_t* p1) {
    bf16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, p0));
    bf16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, p1));
    return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
}

__device__ __forceinline__ bf16x8 pack8(const f32x16& v, int s2) {
    bf16x8 r;
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = (__bf16)v[8 * s2 + j];
    return r;
}

// Register-staged tile copy, split in two halves so that the global loads of tile t+1 are in flight
// while tile t is being consumed (issue early / write late):
//   tile_load : rows [r0, r0+ROWS) x HD of a strided bf16 matrix -> registers (rows >= limit read as zero;
//               addresses are clamped so every load is unconditional and the compiler batches them)
//   tile_store: registers -> LDS tile with the given pitch
template <int HD, int ROWS>
struct TileRegs {
    static constexpr int CPR = HD / 8;                       // 16-B chunks per row
    static constexpr int TOTAL = ROWS * CPR;
    static constexpr int N = (TOTAL + XA_THREADS - 1) / XA_THREADS;
    u32x4 v[N];
};

template <int HD, int ROWS>
__device__ __forceinline__ void tile_load(TileRegs<HD, ROWS>& t, const __bf16* __restrict__ g, int64_t sr, int r0,
                                          int limit) {
    using TR = TileRegs<HD, ROWS>;
#pragma unroll
    for (int i = 0; i < TR::N; ++i) {
        const int c = threadIdx.x + i * XA_THREADS;
        const int row = c / TR::CPR, ch = c % TR::CPR;
        const int rr = min(r0 + row, limit - 1);
        u32x4 v = *(const u32x4*)(g + (int64_t)rr * sr + ch * 8);
        const bool ok = (r0 + row < limit);
        t.v[i] = ok ? v : (u32x4){0u, 0u, 0u, 0u};
    }
}

template <int HD, int ROWS>
__device__ __forceinline__ void tile_store(const TileRegs<HD, ROWS>& t, uint8_t* tile, int pitch) {
    using TR = TileRegs<HD, ROWS>;
#pragma unroll
    for (int i = 0; i < TR::N; ++i) {
        const int c = threadIdx.x + i * XA_THREADS;
        const int row = c / TR::CPR, ch = c % TR::CPR;
        if (TR::TOTAL % XA_THREADS == 0 || c < TR::TOTAL) *(u32x4*)(tile + row * pitch + ch * 16) = t.v[i];
    }
}

// ============================================================================================ forward
template <int HD>
__global__ __launch_bounds__(XA_THREADS) void xattn_fwd_kernel(const __bf16* __restrict__ Q, const __bf16* __restrict__ K,
                                                               const __bf16* __restrict__ V, const float* __restrict__ bias,
                                                               const int32_t* __restrict__ key_len, star_xattn_shape s,
                                                               __bf16* __restrict__ O, float* __restrict__ lse) {
    constexpr int PK = Pitch<HD>::ROW, PV = Pitch<HD>::TR;
    constexpr int KS = HD / 16, NT = HD / 32;
    __shared__ __attribute__((aligned(16))) uint8_t k_lds[KT * PK];
    __shared__ __attribute__((aligned(16))) uint8_t v_lds[KT * PV];
    __shared__ float b_lds[KT];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int r = lane & 31, hf = lane >> 5;
    const int b = blockIdx.z, h = blockIdx.y;
    const int qblk0 = blockIdx.x * 128;
    const int q = qblk0 + wid * 32 + r;            // this lane's query row
    const int qc = min(q, s.L - 1);

    const __bf16* Qb = Q + (int64_t)b * s.q_sb + (int64_t)h * s.q_sh;
    const __bf16* Kb = K + (int64_t)b * s.k_sb + (int64_t)h * s.k_sh;
    const __bf16* Vb = V + (int64_t)b * s.v_sb + (int64_t)h * s.v_sh;

    bf16x8 qf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[ks] = *(const bf16x8*)(Qb + (int64_t)qc * s.q_sr + ks * 16 + hf * 8);

    int kend = s.Kn;
    if (key_len) kend = min(kend, key_len[b]);
    int kloop = kend;                                // keys any query of this workgroup can see
    if (s.ilk_diag >= 0) kloop = min(kloop, min(qblk0 + 128, s.L) + s.ilk_diag);

    f32x16 oacc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int i = 0; i < 16; ++i) oacc[nt][i] = 0.f;
    float m_run = NEG_SENTINEL, l_run = 0.f;

    auto load_bias = [&](int kt0) -> float {
        const int key = kt0 + (tid & (KT - 1));
        return (bias && key < s.bias_len) ? bias[(int64_t)b * s.bias_len + key] : 0.f;
    };
    TileRegs<HD, KT> kreg, vreg;
    float breg = 0.f;
    if (kloop > 0) {
        tile_load<HD, KT>(kreg, Kb, s.k_sr, 0, s.Kn);
        tile_load<HD, KT>(vreg, Vb, s.v_sr, 0, s.Kn);
        breg = load_bias(0);
    }
    for (int kt0 = 0; kt0 < kloop; kt0 += KT) {
        __syncthreads();                       // every wave is done reading the previous tile
        tile_store<HD, KT>(kreg, k_lds, PK);
        tile_store<HD, KT>(vreg, v_lds, PV);
        if (tid < KT) b_lds[tid] = breg;
        __syncthreads();
        if (XA_ABLATE != 4 && kt0 + KT < kloop) {                // next tile's loads fly while this tile is consumed
            tile_load<HD, KT>(kreg, Kb, s.k_sr, kt0 + KT, s.Kn);
            tile_load<HD, KT>(vreg, Vb, s.v_sr, kt0 + KT, s.Kn);
            breg = load_bias(kt0 + KT);
        }

        // S^T[key][q] = sum_n K[key][n] Q[q][n]
        f32x16 sacc[2];
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
            for (int i = 0; i < 16; ++i) sacc[sub][i] = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                bf16x8 kf = lds_row8(k_lds + (sub * 32 + r) * PK + (ks * 2 + hf) * 16);
                if (XA_ABLATE != 3) sacc[sub] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], sacc[sub], 0, 0, 0);
                else asm volatile("" ::"v"(kf));
            }
        }
        // logits, masks, online softmax (the query is on the lane: all row reductions are in-lane)
        float mx = NEG_SENTINEL;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int kl = sub * 32 + crow(i, hf);
                const int key = kt0 + kl;
                float v = sacc[sub][i] * s.scale + b_lds[kl];
                const bool vis = (key < kend) && (s.ilk_diag < 0 || key <= q + s.ilk_diag);
                v = vis ? v : -INFINITY;
                sacc[sub][i] = v;
                mx = fmaxf(mx, v);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = (XA_ABLATE == 1) ? (m_run - m_new) : __expf(m_run - m_new);
        float psum = 0.f;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float p = (XA_ABLATE == 1) ? (sacc[sub][i] - m_new) : __expf(sacc[sub][i] - m_new);
                sacc[sub][i] = p;
                psum += p;
            }
        l_run = l_run * alpha + psum;
        m_run = m_new;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int i = 0; i < 16; ++i) oacc[nt][i] *= alpha;

        // O^T[n][q] += sum_key V[key][n] P[q][key]   (A = V^T via transposed LDS reads, B = P^T accumulator)
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const bf16x8 pf = pack8(sacc[sub], s2);
                const int krow = sub * 32 + s2 * 16 + 4 * hf + ((lane & 15) >> 2);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const int col = nt * 32 + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
                    const uint8_t* p0 = v_lds + krow * PV + col * 2;
                    bf16x8 vf = lds_tr8(p0, p0 + 8 * PV);
                    if (XA_ABLATE != 2) oacc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, oacc[nt], 0, 0, 0);
                    else asm volatile("" ::"v"(vf), "v"(pf));
                }
            }
    }

    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
    if (q < s.L) {
        __bf16* orow = O + (int64_t)b * s.o_sb + (int64_t)h * s.o_sh + (int64_t)q * s.o_sr;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                bf16x4 o4;
#pragma unroll
                for (int j = 0; j < 4; ++j) o4[j] = (__bf16)(oacc[nt][g * 4 + j] * inv);
                *(bf16x4*)(orow + nt * 32 + 8 * g + 4 * hf) = o4;
            }
        if (hf == 0) lse[((int64_t)b * s.h + h) * s.L + q] = l_tot > 0.f ? m_run + __logf(l_tot) : 0.f;
    }
}

// ============================================================================================ delta
// delta[b,h,q] = sum_n dO[q][n] * O[q][n]
template <int HD>
__global__ __launch_bounds__(XA_THREADS) void xattn_delta_kernel(const __bf16* __restrict__ O, const __bf16* __restrict__ dO,
                                                                 star_xattn_shape s, float* __restrict__ delta) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int64_t row = (int64_t)blockIdx.x * 4 + wid;           // flat (b,h,q)
    const int64_t total = (int64_t)s.B * s.h * s.L;
    if (row >= total) return;
    const int q = (int)(row % s.L);
    const int h = (int)((row / s.L) % s.h);
    const int b = (int)(row / ((int64_t)s.L * s.h));
    const int64_t off = (int64_t)b * s.o_sb + (int64_t)h * s.o_sh + (int64_t)q * s.o_sr;
    float acc = 0.f;
    for (int c = lane; c < HD / 2; c += 64) {
        bf16x2 a = *(const bf16x2*)(O + off + c * 2);
        bf16x2 g = *(const bf16x2*)(dO + off + c * 2);
        acc += (float)a[0] * (float)g[0] + (float)a[1] * (float)g[1];
    }
    acc = wave_sum(acc);
    if (lane == 0) delta[row] = acc;
}

// ============================================================================================ dQ
template <int HD>
__global__ __launch_bounds__(XA_THREADS) void xattn_bwd_dq_kernel(
    const __bf16* __restrict__ Q, const __bf16* __restrict__ K, const __bf16* __restrict__ V,
    const __bf16* __restrict__ dO, const float* __restrict__ lse, const float* __restrict__ delta,
    const float* __restrict__ bias, const int32_t* __restrict__ key_len, star_xattn_shape s, __bf16* __restrict__ dQ) {
    constexpr int PK = Pitch<HD>::TR;   // K: row reads (S^T) and transposed reads (dQ^T)
    constexpr int PV = Pitch<HD>::ROW;  // V: row reads only
    constexpr int KS = HD / 16, NT = HD / 32;
    __shared__ __attribute__((aligned(16))) uint8_t k_lds[KT * PK];
    __shared__ __attribute__((aligned(16))) uint8_t v_lds[KT * PV];
    __shared__ float b_lds[KT];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int r = lane & 31, hf = lane >> 5;
    const int b = blockIdx.z, h = blockIdx.y;
    const int qblk0 = blockIdx.x * 128;
    const int q = qblk0 + wid * 32 + r;
    const int qc = min(q, s.L - 1);

    const __bf16* Qb = Q + (int64_t)b * s.q_sb + (int64_t)h * s.q_sh;
    const __bf16* Kb = K + (int64_t)b * s.k_sb + (int64_t)h * s.k_sh;
    const __bf16* Vb = V + (int64_t)b * s.v_sb + (int64_t)h * s.v_sh;
    const __bf16* dOb = dO + (int64_t)b * s.o_sb + (int64_t)h * s.o_sh;

    bf16x8 qf[KS], gf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        qf[ks] = *(const bf16x8*)(Qb + (int64_t)qc * s.q_sr + ks * 16 + hf * 8);
        gf[ks] = *(const bf16x8*)(dOb + (int64_t)qc * s.o_sr + ks * 16 + hf * 8);
    }
    const float my_lse = lse[((int64_t)b * s.h + h) * s.L + qc];
    const float my_delta = delta[((int64_t)b * s.h + h) * s.L + qc];

    int kend = s.Kn;
    if (key_len) kend = min(kend, key_len[b]);
    int kloop = kend;
    if (s.ilk_diag >= 0) kloop = min(kloop, min(qblk0 + 128, s.L) + s.ilk_diag);

    f32x16 dq[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int i = 0; i < 16; ++i) dq[nt][i] = 0.f;

    auto load_bias = [&](int kt0) -> float {
        const int key = kt0 + (tid & (KT - 1));
        return (bias && key < s.bias_len) ? bias[(int64_t)b * s.bias_len + key] : 0.f;
    };
    TileRegs<HD, KT> kreg, vreg;
    float breg = 0.f;
    if (kloop > 0) {
        tile_load<HD, KT>(kreg, Kb, s.k_sr, 0, s.Kn);
        tile_load<HD, KT>(vreg, Vb, s.v_sr, 0, s.Kn);
        breg = load_bias(0);
    }
    for (int kt0 = 0; kt0 < kloop; kt0 += KT) {
        __syncthreads();
        tile_store<HD, KT>(kreg, k_lds, PK);
        tile_store<HD, KT>(vreg, v_lds, PV);
        if (tid < KT) b_lds[tid] = breg;
        __syncthreads();
        if (kt0 + KT < kloop) {
            tile_load<HD, KT>(kreg, Kb, s.k_sr, kt0 + KT, s.Kn);
            tile_load<HD, KT>(vreg, Vb, s.v_sr, kt0 + KT, s.Kn);
            breg = load_bias(kt0 + KT);
        }
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            f32x16 sacc, pacc;
#pragma unroll
            for (int i = 0; i < 16; ++i) { sacc[i] = 0.f; pacc[i] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                bf16x8 kf = lds_row8(k_lds + (sub * 32 + r) * PK + (ks * 2 + hf) * 16);
                sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], sacc, 0, 0, 0);
                bf16x8 vf = lds_row8(v_lds + (sub * 32 + r) * PV + (ks * 2 + hf) * 16);
                pacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, gf[ks], pacc, 0, 0, 0);
            }
            // dS^T = P^T * (dP^T - delta)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int kl = sub * 32 + crow(i, hf);
                const int key = kt0 + kl;
                const bool vis = (key < kend) && (s.ilk_diag < 0 || key <= q + s.ilk_diag);
                const float p = vis ? __expf(sacc[i] * s.scale + b_lds[kl] - my_lse) : 0.f;
                sacc[i] = p * (pacc[i] - my_delta);
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const bf16x8 df = pack8(sacc, s2);
                const int krow = sub * 32 + s2 * 16 + 4 * hf + ((lane & 15) >> 2);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const int col = nt * 32 + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
                    const uint8_t* p0 = k_lds + krow * PK + col * 2;
                    bf16x8 ktf = lds_tr8(p0, p0 + 8 * PK);
                    dq[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ktf, df, dq[nt], 0, 0, 0);
                }
            }
        }
    }
    if (q < s.L) {
        __bf16* drow = dQ + (int64_t)b * s.q_sb + (int64_t)h * s.q_sh + (int64_t)q * s.q_sr;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                bf16x4 o4;
#pragma unroll
                for (int j = 0; j < 4; ++j) o4[j] = (__bf16)(dq[nt][g * 4 + j] * s.scale);
                *(bf16x4*)(drow + nt * 32 + 8 * g + 4 * hf) = o4;
            }
    }
}

// ============================================================================================ dK, dV, dbias
constexpr int KB = 128;   // keys per workgroup (32 per wave)
constexpr int QS = 32;    // query rows per slice

template <int HD>
__global__ __launch_bounds__(XA_THREADS) void xattn_bwd_dkv_kernel(
    const __bf16* __restrict__ Q, const __bf16* __restrict__ K, const __bf16* __restrict__ V,
    const __bf16* __restrict__ dO, const float* __restrict__ lse, const float* __restrict__ delta,
    const float* __restrict__ bias, const int32_t* __restrict__ key_len, star_xattn_shape s, __bf16* __restrict__ dK,
    __bf16* __restrict__ dV, float* __restrict__ dbias_part) {
    constexpr int PR = Pitch<HD>::ROW;  // K, V tiles: row reads
    constexpr int PT = Pitch<HD>::TR;   // Q, dO slices: row reads + transposed reads
    constexpr int KS = HD / 16, NT = HD / 32;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t* k_lds = smem;                       // [KB][PR]
    uint8_t* v_lds = k_lds + KB * PR;            // [KB][PR]
    uint8_t* q_lds = v_lds + KB * PR;            // [QS][PT]
    uint8_t* g_lds = q_lds + QS * PT;            // [QS][PT]
    float* lse_lds = (float*)(g_lds + QS * PT);  // [QS]
    float* del_lds = lse_lds + QS;               // [QS]

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int r = lane & 31, hf = lane >> 5;
    const int b = blockIdx.z, h = blockIdx.y;
    const int kb0 = blockIdx.x * KB;
    const int key = kb0 + wid * 32 + r;          // this lane's key

    const __bf16* Qb = Q + (int64_t)b * s.q_sb + (int64_t)h * s.q_sh;
    const __bf16* Kb = K + (int64_t)b * s.k_sb + (int64_t)h * s.k_sh;
    const __bf16* Vb = V + (int64_t)b * s.v_sb + (int64_t)h * s.v_sh;
    const __bf16* dOb = dO + (int64_t)b * s.o_sb + (int64_t)h * s.o_sh;

    int kend = s.Kn;
    if (key_len) kend = min(kend, key_len[b]);
    const bool key_ok = key < kend;
    const float my_bias = (bias && key < s.bias_len) ? bias[(int64_t)b * s.bias_len + key] : 0.f;

    {
        TileRegs<HD, KB> kreg, vreg;
        tile_load<HD, KB>(kreg, Kb, s.k_sr, kb0, s.Kn);
        tile_load<HD, KB>(vreg, Vb, s.v_sr, kb0, s.Kn);
        tile_store<HD, KB>(kreg, k_lds, PR);
        tile_store<HD, KB>(vreg, v_lds, PR);
    }

    f32x16 dk[NT], dv[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int i = 0; i < 16; ++i) { dk[nt][i] = 0.f; dv[nt][i] = 0.f; }
    float db = 0.f;

    // with the ILK mask key j is seen only by queries i >= j - diag: skip the slices before that
    int q_begin = 0;
    if (s.ilk_diag >= 0) q_begin = max(0, (kb0 - s.ilk_diag) / QS * QS);
    const bool wg_live = kb0 < kend;

    TileRegs<HD, QS> qreg, greg;
    float lreg = 0.f, dreg = 0.f;
    auto load_slice = [&](int q0) {
        tile_load<HD, QS>(qreg, Qb, s.q_sr, q0, s.L);
        tile_load<HD, QS>(greg, dOb, s.o_sr, q0, s.L);
        const int qq = min(q0 + (tid & (QS - 1)), s.L - 1);
        lreg = lse[((int64_t)b * s.h + h) * s.L + qq];
        dreg = delta[((int64_t)b * s.h + h) * s.L + qq];
    };
    if (wg_live && q_begin < s.L) load_slice(q_begin);
    for (int q0 = q_begin; wg_live && q0 < s.L; q0 += QS) {
        __syncthreads();
        tile_store<HD, QS>(qreg, q_lds, PT);
        tile_store<HD, QS>(greg, g_lds, PT);
        if (tid < QS) {
            lse_lds[tid] = lreg;
            del_lds[tid] = dreg;
        }
        __syncthreads();
        if (q0 + QS < s.L) load_slice(q0 + QS);
        // S[q][key] = Q K^T and dP[q][key] = dO V^T  (key on the lane)
        f32x16 sacc, pacc;
#pragma unroll
        for (int i = 0; i < 16; ++i) { sacc[i] = 0.f; pacc[i] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            bf16x8 qa = lds_row8(q_lds + r * PT + (ks * 2 + hf) * 16);
            bf16x8 kf = lds_row8(k_lds + (wid * 32 + r) * PR + (ks * 2 + hf) * 16);
            sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa, kf, sacc, 0, 0, 0);
            bf16x8 ga = lds_row8(g_lds + r * PT + (ks * 2 + hf) * 16);
            bf16x8 vf = lds_row8(v_lds + (wid * 32 + r) * PR + (ks * 2 + hf) * 16);
            pacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ga, vf, pacc, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int ql = crow(i, hf);
            const int qq = q0 + ql;
            const bool vis = key_ok && (qq < s.L) && (s.ilk_diag < 0 || key <= qq + s.ilk_diag);
            const float p = vis ? __expf(sacc[i] * s.scale + my_bias - lse_lds[ql]) : 0.f;
            const float ds = p * (pacc[i] - del_lds[ql]);
            sacc[i] = p;
            pacc[i] = ds;
            db += ds;
        }
        // dV^T[n][key] += sum_q dO[q][n] P[q][key];  dK^T[n][key] += sum_q Q[q][n] dS[q][key]
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const bf16x8 pf = pack8(sacc, s2);
            const bf16x8 df = pack8(pacc, s2);
            const int qrow = s2 * 16 + 4 * hf + ((lane & 15) >> 2);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int col = nt * 32 + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
                const uint8_t* g0 = g_lds + qrow * PT + col * 2;
                bf16x8 gt = lds_tr8(g0, g0 + 8 * PT);
                dv[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gt, pf, dv[nt], 0, 0, 0);
                const uint8_t* q0p = q_lds + qrow * PT + col * 2;
                bf16x8 qt = lds_tr8(q0p, q0p + 8 * PT);
                dk[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qt, df, dk[nt], 0, 0, 0);
            }
        }
    }

    if (key < s.Kn) {
        __bf16* dkrow = dK + (int64_t)b * s.k_sb + (int64_t)h * s.k_sh + (int64_t)key * s.k_sr;
        __bf16* dvrow = dV + (int64_t)b * s.v_sb + (int64_t)h * s.v_sh + (int64_t)key * s.v_sr;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                bf16x4 a, c;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    a[j] = (__bf16)(dk[nt][g * 4 + j] * s.scale);
                    c[j] = (__bf16)dv[nt][g * 4 + j];
                }
                *(bf16x4*)(dkrow + nt * 32 + 8 * g + 4 * hf) = a;
                *(bf16x4*)(dvrow + nt * 32 + 8 * g + 4 * hf) = c;
            }
    }
    db += __shfl_xor(db, 32, 64);
    if (dbias_part && hf == 0 && key < s.Kn) dbias_part[((int64_t)b * s.h + h) * s.Kn + key] = db;
}

// dbias[b,j] (+)= sum_h dbias_part[b,h,j]   (fixed order over heads)
__global__ void xattn_dbias_reduce_kernel(const float* __restrict__ part, int B, int H, int Kn, int bias_len,
                                          int accumulate, float* __restrict__ dbias) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y;
    if (j >= bias_len) return;
    float sum = 0.f;
    for (int h = 0; h < H; ++h) sum += part[((int64_t)b * H + h) * Kn + j];
    float* o = dbias + (int64_t)b * bias_len + j;
    *o = accumulate ? *o + sum : sum;
}

// P[b,h,q,key] = exp(logit - lse): the optional attention-weights output (output_attentions)
template <int HD>
__global__ __launch_bounds__(XA_THREADS) void xattn_probs_kernel(const __bf16* __restrict__ Q, const __bf16* __restrict__ K,
                                                                 const float* __restrict__ bias,
                                                                 const int32_t* __restrict__ key_len,
                                                                 const float* __restrict__ lse, star_xattn_shape s,
                                                                 float* __restrict__ P) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int64_t row = (int64_t)blockIdx.x * 4 + wid;
    const int64_t total = (int64_t)s.B * s.h * s.L;
    if (row >= total) return;
    const int q = (int)(row % s.L);
    const int h = (int)((row / s.L) % s.h);
    const int b = (int)(row / ((int64_t)s.L * s.h));
    const __bf16* qrow = Q + (int64_t)b * s.q_sb + (int64_t)h * s.q_sh + (int64_t)q * s.q_sr;
    int kend = s.Kn;
    if (key_len) kend = min(kend, key_len[b]);
    const float l = lse[row];
    for (int key = lane; key < s.Kn; key += 64) {
        const __bf16* krow = K + (int64_t)b * s.k_sb + (int64_t)h * s.k_sh + (int64_t)key * s.k_sr;
        float acc = 0.f;
        for (int n = 0; n < HD; ++n) acc += (float)qrow[n] * (float)krow[n];
        const bool vis = (key < kend) && (s.ilk_diag < 0 || key <= q + s.ilk_diag);
        const float bj = (bias && key < s.bias_len) ? bias[(int64_t)b * s.bias_len + key] : 0.f;
        P[row * s.Kn + key] = vis ? __expf(acc * s.scale + bj - l) : 0.f;
    }
}

static int check_shape(const star_xattn_shape* s, const char* who) {
    STAR_REQUIRE(s, "%s: null shape", who);
    STAR_REQUIRE(s->B >= 1 && s->h >= 1 && s->L >= 1 && s->Kn >= 1, "%s: bad shape B=%d h=%d L=%d Kn=%d", who, s->B,
                 s->h, s->L, s->Kn);
    STAR_REQUIRE(s->hd == 64 || s->hd == 96 || s->hd == 128, "%s: head dim %d not in {64,96,128}", who, s->hd);
    STAR_REQUIRE(s->B <= 65535 && s->h <= 65535, "%s: B/h exceed grid limits", who);
    STAR_REQUIRE((s->q_sr % 8 == 0) && (s->k_sr % 8 == 0) && (s->v_sr % 8 == 0) && (s->o_sr % 8 == 0) &&
                     (s->q_sb % 8 == 0) && (s->k_sb % 8 == 0) && (s->v_sb % 8 == 0) && (s->o_sb % 8 == 0) &&
                     (s->q_sh % 8 == 0) && (s->k_sh % 8 == 0) && (s->v_sh % 8 == 0) && (s->o_sh % 8 == 0),
                 "%s: strides must be multiples of 8 elements (16-byte rows)", who);
    STAR_REQUIRE(s->bias_len >= 0 && s->bias_len <= s->Kn, "%s: bias_len %d not in [0, Kn=%d]", who, s->bias_len, s->Kn);
    return STAR_OK;
}

template <int HD>
static size_t dkv_lds_bytes() {
    return (size_t)2 * KB * Pitch<HD>::ROW + (size_t)2 * QS * Pitch<HD>::TR + 2 * QS * sizeof(float);
}

template <int HD>
static int launch_fwd(const void* q, const void* k, const void* v, const float* bias, const int32_t* key_len,
                      const star_xattn_shape* s, void* o, float* lse, float* p_out, hipStream_t st) {
    dim3 grid(cdiv(s->L, 128), s->h, s->B);
    hipLaunchKernelGGL(xattn_fwd_kernel<HD>, grid, dim3(XA_THREADS), 0, st, (const __bf16*)q, (const __bf16*)k,
                       (const __bf16*)v, bias, key_len, *s, (__bf16*)o, lse);
    if (p_out) {
        const int64_t rows = (int64_t)s->B * s->h * s->L;
        hipLaunchKernelGGL(xattn_probs_kernel<HD>, dim3(cdiv(rows, 4)), dim3(XA_THREADS), 0, st, (const __bf16*)q,
                           (const __bf16*)k, bias, key_len, (const float*)lse, *s, p_out);
    }
    return STAR_OK;
}

template <int HD>
static int launch_bwd(const void* q, const void* k, const void* v, const void* o, const void* do_, const float* lse,
                      const float* bias, const int32_t* key_len, const star_xattn_shape* s, void* dq, void* dk,
                      void* dv, float* delta, float* dbias_part, float* dbias, int accumulate, hipStream_t st) {
    const int64_t rows = (int64_t)s->B * s->h * s->L;
    hipLaunchKernelGGL(xattn_delta_kernel<HD>, dim3(cdiv(rows, 4)), dim3(XA_THREADS), 0, st, (const __bf16*)o,
                       (const __bf16*)do_, *s, delta);
    hipLaunchKernelGGL(xattn_bwd_dq_kernel<HD>, dim3(cdiv(s->L, 128), s->h, s->B), dim3(XA_THREADS), 0, st,
                       (const __bf16*)q, (const __bf16*)k, (const __bf16*)v, (const __bf16*)do_, lse,
                       (const float*)delta, bias, key_len, *s, (__bf16*)dq);
    const size_t smem = dkv_lds_bytes<HD>();
    hipError_t e = hipFuncSetAttribute((const void*)xattn_bwd_dkv_kernel<HD>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)smem);
    if (e != hipSuccess) {
        set_error("star_xattn_bwd: cannot reserve %zu bytes of LDS: %s", smem, hipGetErrorString(e));
        return STAR_ERR_LAUNCH;
    }
    hipLaunchKernelGGL(xattn_bwd_dkv_kernel<HD>, dim3(cdiv(s->Kn, KB), s->h, s->B), dim3(XA_THREADS), smem, st,
                       (const __bf16*)q, (const __bf16*)k, (const __bf16*)v, (const __bf16*)do_, lse,
                       (const float*)delta, bias, key_len, *s, (__bf16*)dk, (__bf16*)dv, dbias_part);
    if (dbias && s->bias_len > 0)
        hipLaunchKernelGGL(xattn_dbias_reduce_kernel, dim3(cdiv(s->bias_len, 256), s->B), dim3(256), 0, st,
                           (const float*)dbias_part, s->B, s->h, s->Kn, s->bias_len, accumulate, dbias);
    return STAR_OK;
}

}  // namespace star

extern "C" int star_xattn_fwd(const void* q, const void* k, const void* v, const float* bias, const int32_t* key_len,
                              const star_xattn_shape* s, void* o, float* lse, float* p_out, void* stream) {
    using namespace star;
    int rc = check_shape(s, "star_xattn_fwd");
    if (rc) return rc;
    STAR_REQUIRE(q && k && v && o && lse, "star_xattn_fwd: null pointer");
    STAR_REQUIRE(!(bias == nullptr && s->bias_len > 0), "star_xattn_fwd: bias_len > 0 but bias is null");
    hipStream_t st = (hipStream_t)stream;
    switch (s->hd) {
        case 64: rc = launch_fwd<64>(q, k, v, bias, key_len, s, o, lse, p_out, st); break;
        case 96: rc = launch_fwd<96>(q, k, v, bias, key_len, s, o, lse, p_out, st); break;
        default: rc = launch_fwd<128>(q, k, v, bias, key_len, s, o, lse, p_out, st); break;
    }
    if (rc) return rc;
    STAR_LAUNCH_CHECK("star_xattn_fwd");
    return STAR_OK;
}

extern "C" int star_xattn_bwd(const void* q, const void* k, const void* v, const void* o, const void* do_,
                              const float* lse, const float* bias, const int32_t* key_len, const star_xattn_shape* s,
                              void* dq, void* dk, void* dv, float* delta, float* dbias_part, float* dbias,
                              int accumulate, void* stream) {
    using namespace star;
    int rc = check_shape(s, "star_xattn_bwd");
    if (rc) return rc;
    STAR_REQUIRE(q && k && v && o && do_ && lse && dq && dk && dv && delta, "star_xattn_bwd: null pointer");
    STAR_REQUIRE(!(bias == nullptr && s->bias_len > 0), "star_xattn_bwd: bias_len > 0 but bias is null");
    STAR_REQUIRE(!(dbias && !dbias_part), "star_xattn_bwd: dbias requested without dbias_part scratch");
    hipStream_t st = (hipStream_t)stream;
    switch (s->hd) {
        case 64: rc = launch_bwd<64>(q, k, v, o, do_, lse, bias, key_len, s, dq, dk, dv, delta, dbias_part, dbias, accumulate, st); break;
        case 96: rc = launch_bwd<96>(q, k, v, o, do_, lse, bias, key_len, s, dq, dk, dv, delta, dbias_part, dbias, accumulate, st); break;
        default: rc = launch_bwd<128>(q, k, v, o, do_, lse, bias, key_len, s, dq, dk, dv, delta, dbias_part, dbias, accumulate, st); break;
    }
    if (rc) return rc;
    STAR_LAUNCH_CHECK("star_xattn_bwd");
    return STAR_OK;
}
