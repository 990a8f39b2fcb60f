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
//            Short targets (L <= 128, the usual case: ~0.075 tokens per frame) with enough (batch, head) pairs to
//            fill the chip take ONE fused kernel instead (xattn_bwd_fused_small_kernel): dQ, dK, dV and the
//            bias-gradient partials from a single pass over K / V.
//  No atomics anywhere: results are bitwise reproducible.
// Masked logits are -inf (P and dS exactly 0), the running max starts at a finite sentinel.
// Algorithmic FLOP per (b, layer): forward 4*L*k_b*d, backward 8*L*k_b*d (+ 4*L*k_b*d recompute).
#include "common.h"
#include "xattn_drop.h"

namespace star {

constexpr int XA_THREADS = 256;
constexpr int KT = 64;          // keys per LDS tile (forward, dQ)
constexpr float NEG_SENTINEL = -1.0e30f;

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
//   tile_load : rows [r0, r0+ROWS) x HD of a strided bf16 matrix -> registers.  Rows >= limit re-read row
//               limit-1 (clamped address, finite data): every consumer masks them (their P / dS are exact zeros),
//               and a "load or zero" select would make the compiler branch around each load
//   tile_store: registers -> LDS tile with the given pitch
// NTH = threads of the workgroup that take part.
template <int HD, int ROWS, int NTH = XA_THREADS>
struct TileRegs {
    static constexpr int CPR = HD / 8;                       // 16-B chunks per row
    static constexpr int TOTAL = ROWS * CPR;
    static constexpr int N = (TOTAL + NTH - 1) / NTH;
    u32x4 v[N];
};

template <int HD, int ROWS, int NTH>
__device__ __forceinline__ void tile_load(TileRegs<HD, ROWS, NTH>& t, const __bf16* __restrict__ g, int64_t sr, int r0,
                                          int limit) {
    using TR = TileRegs<HD, ROWS, NTH>;
#pragma unroll
    for (int i = 0; i < TR::N; ++i) {
        const int c = threadIdx.x + i * NTH;
        const int row = c / TR::CPR, ch = c % TR::CPR;
        const int rr = min(r0 + row, limit - 1);
        t.v[i] = *(const u32x4*)(g + (int64_t)rr * sr + ch * 8);
    }
}

template <int HD, int ROWS, int NTH>
__device__ __forceinline__ void tile_store(const TileRegs<HD, ROWS, NTH>& t, uint8_t* tile, int pitch) {
    using TR = TileRegs<HD, ROWS, NTH>;
#pragma unroll
    for (int i = 0; i < TR::N; ++i) {
        const int c = threadIdx.x + i * NTH;
        const int row = c / TR::CPR, ch = c % TR::CPR;
        if (TR::TOTAL % NTH == 0 || c < TR::TOTAL) *(u32x4*)(tile + row * pitch + ch * 16) = t.v[i];
    }
}

// ---- in-kernel time stamps (diagnostic builds only: -DSTAR_XA_STAMP; the product library has none)
#ifdef STAR_XA_STAMP
__device__ unsigned long long xa_stamps[1024 * 16];
#define XA_STAMP(slot)                                                                                             \
    do {                                                                                                           \
        if (threadIdx.x == 0)                                                                                      \
            xa_stamps[((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) % 1024 * 16 + (slot)] =     \
                __builtin_amdgcn_s_memrealtime();                                                                  \
    } while (0)
#define XA_STAMP_T(slot, T)                                                                                        \
    do {                                                                                                           \
        if (threadIdx.x == (T))                                                                                    \
            xa_stamps[((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) % 1024 * 16 + (slot)] =     \
                __builtin_amdgcn_s_memrealtime();                                                                  \
    } while (0)
#else
#define XA_STAMP(slot) do {} while (0)
#define XA_STAMP_T(slot, T) do {} while (0)
#endif

constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;
constexpr float RESCALE_THR = 8.0f;   // log2 units: the running max is only raised when a tile exceeds it by 2^8

// ============================================================================================ forward
// One workgroup = 128 queries of one (batch, head): NG groups of 4 waves, 32 queries per wave.  With NG = 2 the
// two groups walk the SAME queries over alternating 64-key tiles (in-workgroup split-K), so every SIMD holds two
// waves whose MFMA / softmax / LDS phases interleave; the partial (m, l, O) states are merged through LDS at the end.
// All softmax arithmetic is in the log2 domain (logit2 = acc * scale*log2e + bias*log2e, p = exp2(logit2 - m2)).
template <int HD, int NG>
struct FwdLds {
    static constexpr int PK = Pitch<HD>::ROW, PV = Pitch<HD>::TR;
    static constexpr int K_BYTES = NG * KT * PK, V_BYTES = NG * KT * PV, B_BYTES = NG * KT * 4;
    static constexpr int TILE_BYTES = K_BYTES + V_BYTES + B_BYTES;
    static constexpr int MERGE_BYTES = (NG > 1) ? ((HD / 32) * 16 + 2) * 256 * 4 : 0;
    static constexpr int BYTES = TILE_BYTES > MERGE_BYTES ? TILE_BYTES : MERGE_BYTES;
};

template <int HD, int NG, bool DROP>
__global__ __launch_bounds__(NG * 256) void xattn_fwd_kernel(
    const __bf16* __restrict__ Q, const __bf16* __restrict__ K, const __bf16* __restrict__ V,
    const float* __restrict__ bias, const int32_t* __restrict__ key_len, star_xattn_shape s, __bf16* __restrict__ O,
    float* __restrict__ lse) {
    constexpr int NTH = NG * 256, QB = 128, KSTEP = NG * KT;
    constexpr int PK = Pitch<HD>::ROW, PV = Pitch<HD>::TR;
    constexpr int KS = HD / 16, NT = HD / 32;
    extern __shared__ __attribute__((aligned(16))) uint8_t fwd_smem[];
    uint8_t* k_all = fwd_smem;
    uint8_t* v_all = k_all + FwdLds<HD, NG>::K_BYTES;
    float* b_all = (float*)(v_all + FwdLds<HD, NG>::V_BYTES);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);     // wave-uniform
    const int grp = wid >> 2, qw = wid & 3;
    const int r = lane & 31, hf = lane >> 5;
    const int b = blockIdx.z, h = blockIdx.y;
    const int qblk0 = blockIdx.x * QB;
    const int q = qblk0 + qw * 32 + r;            // this lane's query row
    const int qc = min(q, s.L - 1);
    const uint8_t* k_lds = k_all + grp * KT * PK;
    const uint8_t* v_lds = v_all + grp * KT * PV;
    const float* b_lds = b_all + grp * KT;

    const __bf16* Qb = Q + (int64_t)b * s.q_sb + (int64_t)h * s.q_sh;
    const __bf16* Kb = K + (int64_t)b * s.k_sb + (int64_t)h * s.k_sh;
    const __bf16* Vb = V + (int64_t)b * s.v_sb + (int64_t)h * s.v_sh;

    bf16x8 qf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[ks] = *(const bf16x8*)(Qb + (int64_t)qc * s.q_sr + ks * 16 + hf * 8);

    const Drop<DROP> drop(s, b, h);
    int kend = s.Kn;
    if (key_len) kend = min(kend, key_len[b]);
    int kloop = kend;                                // keys any query of this workgroup can see
    if (s.ilk_diag >= 0) kloop = min(kloop, min(qblk0 + QB, s.L) + s.ilk_diag);
    const float scale2 = s.scale * LOG2E;

    f32x16 oacc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int i = 0; i < 16; ++i) oacc[nt][i] = 0.f;
    float m_run = NEG_SENTINEL, l_run = 0.f;

    // raw prefetch of the next tile's bias: the value must NOT be touched before the tile is stored to LDS (any
    // arithmetic on it here makes the compiler drain vmcnt(0), i.e. wait for the K / V prefetch issued just before)
    const float* bias_row = (bias && s.bias_len > 0) ? bias + (int64_t)b * s.bias_len : nullptr;
    auto load_bias = [&](int kt0) -> float {
        const int key = kt0 + (tid & (KSTEP - 1));
        return bias_row ? bias_row[min(key, s.bias_len - 1)] : 0.f;
    };
    auto bias2 = [&](float raw, int kt0) -> float {
        const int key = kt0 + (tid & (KSTEP - 1));
        return (key < s.bias_len) ? raw * LOG2E : 0.f;
    };
    TileRegs<HD, KSTEP, NTH> kreg, vreg;
    float breg = 0.f;
    if (kloop > 0) {
        tile_load<HD, KSTEP, NTH>(kreg, Kb, s.k_sr, 0, s.Kn);
        tile_load<HD, KSTEP, NTH>(vreg, Vb, s.v_sr, 0, s.Kn);
        breg = load_bias(0);
    }
    for (int kt0 = 0; kt0 < kloop; kt0 += KSTEP) {
        __syncthreads();                       // every wave is done reading the previous tiles
        tile_store<HD, KSTEP, NTH>(kreg, k_all, PK);
        tile_store<HD, KSTEP, NTH>(vreg, v_all, PV);
        if (tid < KSTEP) b_all[tid] = bias2(breg, kt0);
        __syncthreads();
        if (kt0 + KSTEP < kloop) {             // next tiles' loads fly while these are consumed
            tile_load<HD, KSTEP, NTH>(kreg, Kb, s.k_sr, kt0 + KSTEP, s.Kn);
            tile_load<HD, KSTEP, NTH>(vreg, Vb, s.v_sr, kt0 + KSTEP, s.Kn);
            breg = load_bias(kt0 + KSTEP);
        }
        const int kbase = kt0 + grp * KT;      // first key of this group's tile
        if (kbase >= kloop) continue;          // (wave-uniform) nothing visible in this tile

        // S^T[key][q] = sum_n K[key][n] Q[q][n]
        f32x16 sacc[2];
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
            for (int i = 0; i < 16; ++i) sacc[sub][i] = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                bf16x8 kf = lds_row8(k_lds + (sub * 32 + r) * PK + (ks * 2 + hf) * 16);
                sacc[sub] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], sacc[sub], 0, 0, 0);
            }
        }
        // logits (log2 domain) + bias; the 4 consecutive keys of a register group share one 16-B bias read
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 bb = *(const f32x4*)(b_lds + sub * 32 + 8 * g + 4 * hf);
#pragma unroll
                for (int j = 0; j < 4; ++j) sacc[sub][g * 4 + j] = sacc[sub][g * 4 + j] * scale2 + bb[j];
            }
        // masks only on tiles that need them (wave-uniform test): partial last tile, ILK diagonal
        const bool need_mask = (kbase + KT > kend) || (s.ilk_diag >= 0 && kbase + KT - 1 > qblk0 + qw * 32 + s.ilk_diag);
        if (need_mask) {
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int key = kbase + sub * 32 + crow(i, hf);
                    const bool vis = (key < kend) && (s.ilk_diag < 0 || key <= q + s.ilk_diag);
                    sacc[sub][i] = vis ? sacc[sub][i] : -INFINITY;
                }
        }
        float mx = NEG_SENTINEL;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int i = 0; i < 16; ++i) mx = fmaxf(mx, sacc[sub][i]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        // lazy rescale: raise the running max only when some row of the wave exceeds it by 2^RESCALE_THR
        if (!__all(mx <= m_run + RESCALE_THR)) {
            const float m_new = fmaxf(m_run, mx);
            const float alpha = exp2f(m_run - m_new);
            l_run *= alpha;
            m_run = m_new;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int i = 0; i < 16; ++i) oacc[nt][i] *= alpha;
        }
        float psum = 0.f;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float p = __builtin_amdgcn_exp2f(sacc[sub][i] - m_run);
                psum += p;
                // the softmax denominator is that of the undropped weights; only P V sees the mask
                if constexpr (DROP) sacc[sub][i] = drop.keep(q, kbase + sub * 32 + crow(i, hf)) ? p : 0.f;
                else sacc[sub][i] = p;
            }
        l_run += psum;

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
                    oacc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, oacc[nt], 0, 0, 0);
                }
            }
    }

    if constexpr (NG > 1) {
        // merge the two key groups' partial softmax states (same lane <-> same query / register layout)
        float* mrg = (float*)fwd_smem;
        const int slot = qw * 64 + lane;
        __syncthreads();
        if (grp == 1) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int i = 0; i < 16; ++i) mrg[(nt * 16 + i) * 256 + slot] = oacc[nt][i];
            mrg[(NT * 16) * 256 + slot] = m_run;
            mrg[(NT * 16 + 1) * 256 + slot] = l_run;
        }
        __syncthreads();
        if (grp == 1) return;
        const float m1 = mrg[(NT * 16) * 256 + slot], l1 = mrg[(NT * 16 + 1) * 256 + slot];
        const float m = fmaxf(m_run, m1);
        const float a0 = exp2f(m_run - m), a1 = exp2f(m1 - m);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int i = 0; i < 16; ++i) oacc[nt][i] = oacc[nt][i] * a0 + mrg[(nt * 16 + i) * 256 + slot] * a1;
        l_run = l_run * a0 + l1 * a1;
        m_run = m;
    }

    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
    if constexpr (DROP) inv *= drop.inv_keep;
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
        // natural-log LSE of the masked, biased logits
        if (hf == 0) lse[((int64_t)b * s.h + h) * s.L + q] = l_tot > 0.f ? (m_run + __log2f(l_tot)) * LN2 : 0.f;
    }
}

// ============================================================================================ dQ (+ delta)
// Same walk as the forward (query on the lane, NG key groups merged through LDS).  delta = rowsum(dO*O) is
// computed here from the fragments the kernel holds anyway and written out for the dK/dV kernel.
template <int HD, int NG>
struct DqLds {
    static constexpr int PK = Pitch<HD>::TR, PV = Pitch<HD>::ROW;
    static constexpr int K_BYTES = NG * KT * PK, V_BYTES = NG * KT * PV, B_BYTES = NG * KT * 4;
    static constexpr int TILE_BYTES = K_BYTES + V_BYTES + B_BYTES;
    static constexpr int MERGE_BYTES = (NG > 1) ? (HD / 32) * 16 * 256 * 4 : 0;
    static constexpr int BYTES = TILE_BYTES > MERGE_BYTES ? TILE_BYTES : MERGE_BYTES;
};

template <int HD, int NG, bool DROP>
__global__ __launch_bounds__(NG * 256) void xattn_bwd_dq_kernel(
    const __bf16* __restrict__ Q, const __bf16* __restrict__ K, const __bf16* __restrict__ V,
    const __bf16* __restrict__ O, const __bf16* __restrict__ dO, const float* __restrict__ lse,
    float* __restrict__ delta, const float* __restrict__ bias, const int32_t* __restrict__ key_len,
    star_xattn_shape s, __bf16* __restrict__ dQ) {
    constexpr int NTH = NG * 256, QB = 128, KSTEP = NG * KT;
    constexpr int PK = Pitch<HD>::TR;   // K: row reads (S^T) and transposed reads (dQ^T)
    constexpr int PV = Pitch<HD>::ROW;  // V: row reads only
    constexpr int KS = HD / 16, NT = HD / 32;
    extern __shared__ __attribute__((aligned(16))) uint8_t dq_smem[];
    uint8_t* k_all = dq_smem;
    uint8_t* v_all = k_all + DqLds<HD, NG>::K_BYTES;
    float* b_all = (float*)(v_all + DqLds<HD, NG>::V_BYTES);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);     // wave-uniform
    const int grp = wid >> 2, qw = wid & 3;
    const int r = lane & 31, hf = lane >> 5;
    const int b = blockIdx.z, h = blockIdx.y;
    const int qblk0 = blockIdx.x * QB;
    const int q = qblk0 + qw * 32 + r;
    const int qc = min(q, s.L - 1);
    const uint8_t* k_lds = k_all + grp * KT * PK;
    const uint8_t* v_lds = v_all + grp * KT * PV;
    const float* b_lds = b_all + grp * KT;

    const __bf16* Qb = Q + (int64_t)b * s.q_sb + (int64_t)h * s.q_sh;
    const __bf16* Kb = K + (int64_t)b * s.k_sb + (int64_t)h * s.k_sh;
    const __bf16* Vb = V + (int64_t)b * s.v_sb + (int64_t)h * s.v_sh;
    const __bf16* Ob = O + (int64_t)b * s.o_sb + (int64_t)h * s.o_sh;
    const __bf16* dOb = dO + (int64_t)b * s.o_sb + (int64_t)h * s.o_sh;

    bf16x8 qf[KS], gf[KS];
    float my_delta = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        qf[ks] = *(const bf16x8*)(Qb + (int64_t)qc * s.q_sr + ks * 16 + hf * 8);
        gf[ks] = *(const bf16x8*)(dOb + (int64_t)qc * s.o_sr + ks * 16 + hf * 8);
        const bf16x8 of = *(const bf16x8*)(Ob + (int64_t)qc * s.o_sr + ks * 16 + hf * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) my_delta += (float)gf[ks][j] * (float)of[j];
    }
    my_delta += __shfl_xor(my_delta, 32, 64);
    if (grp == 0 && hf == 0 && q < s.L) delta[((int64_t)b * s.h + h) * s.L + q] = my_delta;
    const float my_lse2 = lse[((int64_t)b * s.h + h) * s.L + qc] * LOG2E;

    const Drop<DROP> drop(s, b, h);
    int kend = s.Kn;
    if (key_len) kend = min(kend, key_len[b]);
    int kloop = kend;
    if (s.ilk_diag >= 0) kloop = min(kloop, min(qblk0 + QB, s.L) + s.ilk_diag);
    const float scale2 = s.scale * LOG2E;

    f32x16 dq[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int i = 0; i < 16; ++i) dq[nt][i] = 0.f;

    // raw prefetch of the next tile's bias: the value must NOT be touched before the tile is stored to LDS (any
    // arithmetic on it here makes the compiler drain vmcnt(0), i.e. wait for the K / V prefetch issued just before)
    const float* bias_row = (bias && s.bias_len > 0) ? bias + (int64_t)b * s.bias_len : nullptr;
    auto load_bias = [&](int kt0) -> float {
        const int key = kt0 + (tid & (KSTEP - 1));
        return bias_row ? bias_row[min(key, s.bias_len - 1)] : 0.f;
    };
    auto bias2 = [&](float raw, int kt0) -> float {
        const int key = kt0 + (tid & (KSTEP - 1));
        return (key < s.bias_len) ? raw * LOG2E : 0.f;
    };
    TileRegs<HD, KSTEP, NTH> kreg, vreg;
    float breg = 0.f;
    if (kloop > 0) {
        tile_load<HD, KSTEP, NTH>(kreg, Kb, s.k_sr, 0, s.Kn);
        tile_load<HD, KSTEP, NTH>(vreg, Vb, s.v_sr, 0, s.Kn);
        breg = load_bias(0);
    }
    for (int kt0 = 0; kt0 < kloop; kt0 += KSTEP) {
        __syncthreads();
        tile_store<HD, KSTEP, NTH>(kreg, k_all, PK);
        tile_store<HD, KSTEP, NTH>(vreg, v_all, PV);
        if (tid < KSTEP) b_all[tid] = bias2(breg, kt0);
        __syncthreads();
        if (kt0 + KSTEP < kloop) {
            tile_load<HD, KSTEP, NTH>(kreg, Kb, s.k_sr, kt0 + KSTEP, s.Kn);
            tile_load<HD, KSTEP, NTH>(vreg, Vb, s.v_sr, kt0 + KSTEP, s.Kn);
            breg = load_bias(kt0 + KSTEP);
        }
        const int kbase = kt0 + grp * KT;
        if (kbase >= kloop) continue;
        const bool need_mask = (kbase + KT > kend) || (s.ilk_diag >= 0 && kbase + KT - 1 > qblk0 + qw * 32 + s.ilk_diag);
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
            // P^T = exp2(logit2 - lse2);  dS^T = P^T * (dP^T - delta)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 bb = *(const f32x4*)(b_lds + sub * 32 + 8 * g + 4 * hf);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int i = g * 4 + j;
                    float p = __builtin_amdgcn_exp2f(sacc[i] * scale2 + bb[j] - my_lse2);
                    if constexpr (DROP) sacc[i] = p * (pacc[i] * drop.km(q, kbase + sub * 32 + crow(i, hf)) - my_delta);
                    else sacc[i] = p * (pacc[i] - my_delta);
                }
            }
            if (need_mask) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int key = kbase + sub * 32 + crow(i, hf);
                    const bool vis = (key < kend) && (s.ilk_diag < 0 || key <= q + s.ilk_diag);
                    sacc[i] = vis ? sacc[i] : 0.f;
                }
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
    if constexpr (NG > 1) {
        float* mrg = (float*)dq_smem;
        const int slot = qw * 64 + lane;
        __syncthreads();
        if (grp == 1) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int i = 0; i < 16; ++i) mrg[(nt * 16 + i) * 256 + slot] = dq[nt][i];
        }
        __syncthreads();
        if (grp == 1) return;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int i = 0; i < 16; ++i) dq[nt][i] += mrg[(nt * 16 + i) * 256 + slot];
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

// ---- LDS-DMA ring variant of the dQ kernel (HD = 64 or 128).  One workgroup = 8 waves x 16 queries.  K / V /
// bias tiles of 64 keys are streamed by global_load_lds (lane-linear LDS image, XOR swizzle on the per-lane
// SOURCE address and on the read address) into a 4-stage ring: ONE raw barrier per tile, counted vmcnt, two
// tiles in flight across every barrier, so HBM stays busy from the first cycle to the last.  16x16x32 MFMAs,
// query on the MFMA column: S^T = K Q^T and dP^T = V dO^T (A rows from LDS, B = Q / dO fragments in registers),
// dS^T is packed in-lane into the B operand of dQ^T = K^T dS^T (A by ds_read_b64_tr_b16 of the same K tile).
template <int HD>
struct Ring {
    static constexpr int KT2 = 64, NST = 4, WAVES = 8;
    static constexpr int ROWB = HD * 2;
    static constexpr int TILE = KT2 * ROWB;                 // one operand tile
    static constexpr int STAGE = 2 * TILE + KT2 * 4;        // K | V | bias
    static constexpr int BYTES = NST * STAGE;
    static constexpr int PIECES = TILE / 1024 / WAVES;      // 1-KiB DMA pieces per wave per operand tile
    static constexpr int RPP = 1024 / ROWB;                 // rows per piece
    static_assert(PIECES >= 1, "tile too small for 8 waves");
    __device__ static __forceinline__ int swz(int r) {
        if constexpr (HD == 128) return (r & 7) << 1;
        else return (((r >> 1) & 1) << 2) | (((r >> 2) & 1) << 1) | ((r >> 2) & 1);
    }
};

__device__ __forceinline__ void xa_glds16(const void* g, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
__device__ __forceinline__ void xa_glds4(const void* g, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 4, 0, 0);
}
// wait until at most n of this wave's vector-memory operations are outstanding (n is wave-uniform; the immediate of
// s_waitcnt must be a constant, hence the switch; counts above 32 wait for 32, which is only ever stricter)
__device__ __forceinline__ void xa_wait_vm(int n) {
    switch (n > 32 ? 32 : n) {
        case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
        case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
        case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
        case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
        case 13: asm volatile("s_waitcnt vmcnt(13)" ::: "memory"); break;
        case 14: asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); break;
        case 15: asm volatile("s_waitcnt vmcnt(15)" ::: "memory"); break;
        case 16: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
        case 17: asm volatile("s_waitcnt vmcnt(17)" ::: "memory"); break;
        case 18: asm volatile("s_waitcnt vmcnt(18)" ::: "memory"); break;
        case 19: asm volatile("s_waitcnt vmcnt(19)" ::: "memory"); break;
        case 20: asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); break;
        case 21: asm volatile("s_waitcnt vmcnt(21)" ::: "memory"); break;
        case 22: asm volatile("s_waitcnt vmcnt(22)" ::: "memory"); break;
        case 23: asm volatile("s_waitcnt vmcnt(23)" ::: "memory"); break;
        case 24: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
        case 25: asm volatile("s_waitcnt vmcnt(25)" ::: "memory"); break;
        case 26: asm volatile("s_waitcnt vmcnt(26)" ::: "memory"); break;
        case 27: asm volatile("s_waitcnt vmcnt(27)" ::: "memory"); break;
        case 28: asm volatile("s_waitcnt vmcnt(28)" ::: "memory"); break;
        case 29: asm volatile("s_waitcnt vmcnt(29)" ::: "memory"); break;
        case 30: asm volatile("s_waitcnt vmcnt(30)" ::: "memory"); break;
        case 31: asm volatile("s_waitcnt vmcnt(31)" ::: "memory"); break;
        case 32: asm volatile("s_waitcnt vmcnt(32)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

// Transposed LDS read as inline asm.  The compiler has no memory operand for the ds_read_tr builtin, so next to an
// LDS-DMA in flight it drains vmcnt(0) before every one of them (which serialises the ring); an asm read is not
// tracked at all: the caller waits with xa_wait_lds() before the first use of the results.
__device__ __forceinline__ uint32_t lds_addr(const void* p) {
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)p;
}
template <int OFF2>
__device__ __forceinline__ bf16x8 lds_tr8_async(uint32_t addr) {
    bf16x4 a, b;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(a) : "v"(addr));
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(b) : "v"(addr), "n"(OFF2));
    return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
}
__device__ __forceinline__ void xa_wait_lds() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

// Register loads that the compiler does not track (so that it does not drain the LDS-DMA queue with vmcnt(0) at
// their first use): the caller counts them and waits with xa_wait_vm() + sched_barrier before using the values.
// (macros, not functions: the asm writes the destination variable itself, so no register copy of a value that
// has not arrived yet can be scheduled between the load and the wait)
#define XA_GLOAD16_ASYNC(dst, ptr) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(ptr) : "memory")
#define XA_GLOAD4_ASYNC(dst, ptr) asm volatile("global_load_dword %0, %1, off" : "=v"(dst) : "v"(ptr) : "memory")

// issue the DMA of K / V / bias tile `tile` (keys tile*64 ..) into ring slot tile % NST
template <int HD, int NW = 8>
__device__ __forceinline__ void ring_stage(uint8_t* smem, const __bf16* Kb, int64_t k_sr, const __bf16* Vb, int64_t v_sr,
                                           const float* bias_row, int bias_len, int Kn, int tile, int wid, int lane) {
    using R = Ring<HD>;
    constexpr int PIECES = R::TILE / 1024 / NW;      // 1-KiB DMA pieces per loading wave per operand tile
    static_assert(PIECES >= 1, "tile too small for this many loader waves");
    uint8_t* st = smem + (tile % R::NST) * R::STAGE;
    const int key0 = tile * R::KT2;
#pragma unroll
    for (int i = 0; i < PIECES; ++i) {
        const int piece = wid * PIECES + i;
        const int row = piece * R::RPP + lane / (R::ROWB / 16);
        const int pos = lane % (R::ROWB / 16);
        const int chunk = pos ^ R::swz(row & 15);
        const int key = min(key0 + row, Kn - 1);
        xa_glds16(Kb + (int64_t)key * k_sr + chunk * 8, st + piece * 1024);
        xa_glds16(Vb + (int64_t)key * v_sr + chunk * 8, st + R::TILE + piece * 1024);
    }
    if (wid == 0) {
        // bias[key] for the 64 keys of the tile (keys past bias_len read a valid address and are zeroed at use)
        const int key = min(key0 + lane, max(bias_len, 1) - 1);
        xa_glds4(bias_row + key, st + 2 * R::TILE);
    }
}

// ---- forward on the same ring (HD = 64 or 128).  The register-staged forward above keeps ONE K / V step in flight and
// every workgroup of the launch walks its steps in lock step (one workgroup per CU at the usual shapes), so HBM idles
// while the chip computes and the other way round; and an LDS-DMA piece costs the wave that issues it 100-200 issue
// cycles, which a wave that also computes pays on its critical path.  Here the roles are split: 8 waves =
//   4 consumer waves x 32 queries = two 16-query blocks that share every K / V fragment read from LDS (LDS read
//     bandwidth is what bounds the tile loop: eight waves x 16 queries re-read each tile twice as often and took
//     1.2 us per 64 keys) (16x16x32 MFMAs, query on the MFMA column): S^T = K Q^T (A rows from LDS, B = Q
//     fragments in registers); online softmax in the log2 domain with a lazily raised running max (the usual tile
//     needs neither a cross-lane max nor a rescale); the bf16 P^T block is packed in-lane into the B operand of
//     O^T = V^T P^T (A by ds_read_b64_tr_b16 of the V tile, same key permutation on both sides); no merge step;
//   4 loader waves (one per SIMD) that stream the K / V / bias tiles of 64 keys through the 4-slot ring: four tiles
//     (128 KiB at HD = 128, more than half of a 375-key memory) are requested before the first MFMA and three stay in
//     flight behind every barrier.
// One raw barrier per tile joins them: a loader arrives once its pieces of the tile have landed (counted vmcnt), a
// consumer once it has finished the previous tile, whose slot the loaders refill right after the barrier.
constexpr int FR_CONSUMERS = 4, FR_LOADERS = 4, FR_THREADS = (FR_CONSUMERS + FR_LOADERS) * 64;
constexpr int FR_NQ = 2;          // 16-query blocks per consumer wave (they share every K / V fragment read from LDS)

template <int HD, bool HAS_BIAS, bool DROP>
__global__ __launch_bounds__(FR_THREADS) void xattn_fwd_ring_kernel(
    const __bf16* __restrict__ Q, const __bf16* __restrict__ K, const __bf16* __restrict__ V,
    const float* __restrict__ bias, const float* __restrict__ zero_bias, const int32_t* __restrict__ key_len,
    star_xattn_shape s, __bf16* __restrict__ O, float* __restrict__ lse) {
    using R = Ring<HD>;
    constexpr int ROWB = R::ROWB, KT2 = R::KT2;
    constexpr int KS = HD / 32;       // k-steps of a 16x16x32 MFMA over the head dim
    constexpr int NT = HD / 16;       // 16-wide head-dim tiles of O^T
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.z, h = blockIdx.y;

    int kend = s.Kn;
    if (key_len) kend = min(kend, key_len[b]);
    int kloop_wg = kend;                             // keys the workgroup has to visit
    if (s.ilk_diag >= 0) kloop_wg = min(kend, min((int)blockIdx.x * 128 + 128, s.L) + s.ilk_diag);
    const int ntiles = (kloop_wg + KT2 - 1) / KT2;
    XA_STAMP(0);
#ifdef STAR_XA_STAMP
    if (threadIdx.x == 0) xa_stamps[((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) % 1024 * 16 + 7] = __builtin_amdgcn_s_memtime();
#endif

    if (wid >= FR_CONSUMERS) {
        // ------------------------------------------------------------------ loader waves
        const int lw = wid - FR_CONSUMERS;
        const __bf16* Kb = K + (int64_t)b * s.k_sb + (int64_t)h * s.k_sh;
        const __bf16* Vb = V + (int64_t)b * s.v_sb + (int64_t)h * s.v_sh;
        const float* bias_row = HAS_BIAS ? bias + (int64_t)b * s.bias_len : zero_bias;
        const int bias_len = HAS_BIAS ? s.bias_len : 0;
        const int ppt = 2 * (R::TILE / 1024 / FR_LOADERS) + (lw == 0 ? 1 : 0);      // DMA operations of this wave per tile
        XA_STAMP_T(8, FR_CONSUMERS * 64);
        // tiles 0 and 1 first; the first barrier falls as soon as tile 0 has landed (the first DMAs of a launch are
        // slow to issue -- cold address translation -- so filling all four slots first would hold the consumers back)
#pragma unroll 1
        for (int t = 0; t < min(ntiles, 2); ++t) {
            ring_stage<HD, FR_LOADERS>(smem, Kb, s.k_sr, Vb, s.v_sr, bias_row, bias_len, s.Kn, t, lw, lane);
            XA_STAMP_T(9 + t, FR_CONSUMERS * 64);
        }
#pragma unroll 1
        for (int tile = 0; tile < ntiles; ++tile) {
            // tile `tile` has landed once only the DMAs of the younger tiles are outstanding: tile 1 at tile 0, two
            // tiles afterwards (2 and 3 are issued behind the first barrier, tile + 3 behind every later one)
            xa_wait_vm(ppt * min(tile == 0 ? 1 : 2, ntiles - 1 - tile));
            if (tile == 0) XA_STAMP_T(13, FR_CONSUMERS * 64);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            // slot (tile+3) % 4 held tile-1, which every consumer finished before the barrier above
            if (tile == 0) {
                if (ntiles > 2) ring_stage<HD, FR_LOADERS>(smem, Kb, s.k_sr, Vb, s.v_sr, bias_row, bias_len, s.Kn, 2, lw, lane);
                if (ntiles > 3) ring_stage<HD, FR_LOADERS>(smem, Kb, s.k_sr, Vb, s.v_sr, bias_row, bias_len, s.Kn, 3, lw, lane);
            } else if (tile + 3 < ntiles) {
                ring_stage<HD, FR_LOADERS>(smem, Kb, s.k_sr, Vb, s.v_sr, bias_row, bias_len, s.Kn, tile + 3, lw, lane);
            }
        }
        return;
    }

    // ---------------------------------------------------------------------- consumer waves
    const int r16 = lane & 15, g = lane >> 4;
    const int q0 = blockIdx.x * 128 + wid * (16 * FR_NQ);      // first query of this wave
    const int bias_len = HAS_BIAS ? s.bias_len : 0;
    const Drop<DROP> drop(s, b, h);
    int kloop = kend;                                // keys this wave has to visit
    if (s.ilk_diag >= 0) kloop = min(kend, min(q0 + 16 * FR_NQ, s.L) + s.ilk_diag);
    if (q0 >= s.L) kloop = 0;

    bf16x8 qf[FR_NQ][KS];
    int qi[FR_NQ], klim[FR_NQ], klim_wave[FR_NQ];
#pragma unroll
    for (int qb = 0; qb < FR_NQ; ++qb) {
        const int q = q0 + 16 * qb + r16;
        qi[qb] = q;
        const __bf16* qrow = Q + (int64_t)b * s.q_sb + (int64_t)h * s.q_sh + (int64_t)min(q, s.L - 1) * s.q_sr + g * 8;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) qf[qb][ks] = *(const bf16x8*)(qrow + ks * 32);
        // per-lane visibility limit: key visible  <=>  key < klim
        int kl = kend;
        if (s.ilk_diag >= 0) kl = min(kl, q + s.ilk_diag + 1);
        if (q >= s.L) kl = 0;
        klim[qb] = kl;
        // every lane of the block sees all keys below klim_wave (0: rows past L -> always the masked path)
        int kw = kend;
        if (s.ilk_diag >= 0) kw = min(kw, q0 + 16 * qb + s.ilk_diag + 1);
        if (q0 + 16 * qb + 16 > s.L) kw = 0;
        klim_wave[qb] = kw;
    }
    const float scale2 = s.scale * LOG2E;

    f32x4 oacc[FR_NQ][NT];
    float m_run[FR_NQ], l_run[FR_NQ];
#pragma unroll
    for (int qb = 0; qb < FR_NQ; ++qb) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) oacc[qb][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        m_run[qb] = NEG_SENTINEL;
        l_run[qb] = 0.f;
    }

    const int sw_row = R::swz(r16);
    const int tr_row = 4 * g + (r16 >> 2);
    const int sw_tr = R::swz(tr_row);

#pragma unroll 1
    for (int tile = 0; tile < ntiles; ++tile) {
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        XA_STAMP(1 + tile);
        const uint8_t* k_lds = smem + (tile % R::NST) * R::STAGE;
        const uint8_t* v_lds = k_lds + R::TILE;
        const float* b_lds = (const float*)(k_lds + 2 * R::TILE);
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            const int kb0 = tile * KT2 + sub * 32;
            if (kb0 >= kloop) break;                  // wave-uniform
            bf16x8 ka[2][KS];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int rowoff = (sub * 32 + 16 * t + r16) * ROWB;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) ka[t][ks] = lds_row8(k_lds + rowoff + (((4 * ks + g) ^ sw_row) << 4));
            }
            f32x4 bb[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) bb[t] = *(const f32x4*)(b_lds + sub * 32 + 16 * t + 4 * g);
            // transposed V fragments: in flight during the S^T products and the softmax arithmetic
            const uint8_t* trp = v_lds + (sub * 32 + tr_row) * ROWB + 8 * (lane & 1);
            bf16x8 vt[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const uint8_t* p0 = trp + (((2 * nt + ((lane & 3) >> 1)) ^ sw_tr) << 4);
                vt[nt] = lds_tr8(p0, p0 + 16 * ROWB);
            }
            f32x4 sacc[FR_NQ][2];
#pragma unroll
            for (int qb = 0; qb < FR_NQ; ++qb)
#pragma unroll
                for (int t = 0; t < 2; ++t) sacc[qb][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int qb = 0; qb < FR_NQ; ++qb)
                        sacc[qb][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ka[t][ks], qf[qb][ks], sacc[qb][t], 0, 0, 0);
            bf16x8 pf[FR_NQ];
#pragma unroll
            for (int qb = 0; qb < FR_NQ; ++qb) {
                // logits in the log2 domain; masks only where a step needs them
                const bool plain = (kb0 + 32 <= klim_wave[qb]) && (!HAS_BIAS || kb0 + 32 <= bias_len);    // wave-uniform
                float lg[8];
                if (plain) {
#pragma unroll
                    for (int t = 0; t < 2; ++t)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            lg[4 * t + j] = HAS_BIAS ? __builtin_fmaf(sacc[qb][t][j], scale2, bb[t][j] * LOG2E)
                                                     : sacc[qb][t][j] * scale2;
                } else {
#pragma unroll
                    for (int t = 0; t < 2; ++t)
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int key = kb0 + 16 * t + 4 * g + j;
                            const float bj = (HAS_BIAS && key < bias_len) ? bb[t][j] * LOG2E : 0.f;
                            const float v = sacc[qb][t][j] * scale2 + bj;
                            lg[4 * t + j] = (key < klim[qb]) ? v : -INFINITY;
                        }
                }
                float mx = NEG_SENTINEL;
#pragma unroll
                for (int i = 0; i < 8; ++i) mx = fmaxf(mx, lg[i]);
                // lazy rescale: the running max (common to the four lane groups of a query) is raised only when some
                // lane exceeds it by 2^RESCALE_THR -- the usual step needs neither a cross-lane max nor a rescale
                if (!__all(mx <= m_run[qb] + RESCALE_THR)) {
                    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
                    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                    const float m_new = fmaxf(m_run[qb], mx);
                    const float alpha = exp2f(m_run[qb] - m_new);
                    l_run[qb] *= alpha;
                    m_run[qb] = m_new;
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                        for (int j = 0; j < 4; ++j) oacc[qb][nt][j] *= alpha;
                }
                float psum = 0.f;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    float p = __builtin_amdgcn_exp2f(lg[i] - m_run[qb]);
                    psum += p;
                    // the softmax denominator is that of the undropped weights; only P V sees the mask
                    if constexpr (DROP) p = drop.keep(qi[qb], kb0 + 16 * (i >> 2) + 4 * g + (i & 3)) ? p : 0.f;
                    pf[qb][i] = (__bf16)p;
                }
                l_run[qb] += psum;
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int qb = 0; qb < FR_NQ; ++qb)
                    oacc[qb][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vt[nt], pf[qb], oacc[qb][nt], 0, 0, 0);
        }
    }

    XA_STAMP(14);
#pragma unroll
    for (int qb = 0; qb < FR_NQ; ++qb) {
        float l_tot = l_run[qb] + __shfl_xor(l_run[qb], 16, 64);
        l_tot += __shfl_xor(l_tot, 32, 64);
        float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
        if constexpr (DROP) inv *= drop.inv_keep;
        const int q = qi[qb];
        if (q < s.L) {
            __bf16* orow = O + (int64_t)b * s.o_sb + (int64_t)h * s.o_sh + (int64_t)q * s.o_sr + 4 * g;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                bf16x4 o4;
#pragma unroll
                for (int j = 0; j < 4; ++j) o4[j] = (__bf16)(oacc[qb][nt][j] * inv);
                *(bf16x4*)(orow + nt * 16) = o4;
            }
            // natural-log LSE of the masked, biased logits
            if (g == 0) lse[((int64_t)b * s.h + h) * s.L + q] = l_tot > 0.f ? (m_run[qb] + __log2f(l_tot)) * LN2 : 0.f;
        }
    }
#ifdef STAR_XA_STAMP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    XA_STAMP(15);
    if (threadIdx.x == 0) xa_stamps[((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) % 1024 * 16 + 12] = __builtin_amdgcn_s_memtime();
#endif
}

template <int HD, bool HAS_BIAS, bool DROP>
__global__ __launch_bounds__(512) void xattn_bwd_dq_ring_kernel(
    const __bf16* __restrict__ Q, const __bf16* __restrict__ K, const __bf16* __restrict__ V,
    const __bf16* __restrict__ O, const __bf16* __restrict__ dO, const float* __restrict__ lse,
    float* __restrict__ delta, const float* __restrict__ bias, const float* __restrict__ zero_bias,
    const int32_t* __restrict__ key_len, star_xattn_shape s, __bf16* __restrict__ dQ) {
    using R = Ring<HD>;
    constexpr int ROWB = R::ROWB, KT2 = R::KT2;
    constexpr int KS = HD / 32;       // k-steps of a 16x16x32 MFMA over the head dim
    constexpr int NT = HD / 16;       // 16-wide head-dim tiles of dQ^T
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, g = lane >> 4;
    const int b = blockIdx.z, h = blockIdx.y;
    const int q0 = blockIdx.x * 128 + wid * 16;      // first query of this wave
    const int q = q0 + r16;
    const int qc = min(q, s.L - 1);

    const __bf16* Kb = K + (int64_t)b * s.k_sb + (int64_t)h * s.k_sh;
    const __bf16* Vb = V + (int64_t)b * s.v_sb + (int64_t)h * s.v_sh;
    const float* bias_row = HAS_BIAS ? bias + (int64_t)b * s.bias_len : zero_bias;
    const int bias_len = HAS_BIAS ? s.bias_len : 0;
    const Drop<DROP> drop(s, b, h);

    int kend = s.Kn;
    if (key_len) kend = min(kend, key_len[b]);
    // keys the workgroup / this wave have to visit
    int kloop_wg = kend, kloop = kend;
    if (s.ilk_diag >= 0) {
        kloop_wg = min(kend, min((int)blockIdx.x * 128 + 128, s.L) + s.ilk_diag);
        kloop = min(kend, min(q0 + 16, s.L) + s.ilk_diag);
    }
    if (q0 >= s.L) kloop = 0;
    const int ntiles = (kloop_wg + KT2 - 1) / KT2;
    const int ppt = 2 * R::PIECES + (wid == 0 ? 1 : 0);      // DMA operations of this wave per tile

    // tile 0 first, then the register operands (untracked asm loads), then tiles 1 and 2
    if (ntiles > 0) ring_stage<HD>(smem, Kb, s.k_sr, Vb, s.v_sr, bias_row, bias_len, s.Kn, 0, wid, lane);

    const __bf16* qrow = Q + (int64_t)b * s.q_sb + (int64_t)h * s.q_sh + (int64_t)qc * s.q_sr + g * 8;
    const __bf16* orow = O + (int64_t)b * s.o_sb + (int64_t)h * s.o_sh + (int64_t)qc * s.o_sr + g * 8;
    const __bf16* grow = dO + (int64_t)b * s.o_sb + (int64_t)h * s.o_sh + (int64_t)qc * s.o_sr + g * 8;
    bf16x8 qf[KS], gf[KS], of[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        XA_GLOAD16_ASYNC(qf[ks], qrow + ks * 32);
        XA_GLOAD16_ASYNC(gf[ks], grow + ks * 32);
        XA_GLOAD16_ASYNC(of[ks], orow + ks * 32);
    }
    float my_lse2;
    XA_GLOAD4_ASYNC(my_lse2, lse + ((int64_t)b * s.h + h) * s.L + qc);
    // tiles 1 and 2 follow at once, so HBM stays busy; then wait for everything older than them
    if (ntiles > 1) ring_stage<HD>(smem, Kb, s.k_sr, Vb, s.v_sr, bias_row, bias_len, s.Kn, 1, wid, lane);
    if (ntiles > 2) ring_stage<HD>(smem, Kb, s.k_sr, Vb, s.v_sr, bias_row, bias_len, s.Kn, 2, wid, lane);
    xa_wait_vm(ppt * min(2, max(ntiles - 1, 0)));
    __builtin_amdgcn_sched_barrier(0);
    float my_delta = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j) my_delta += (float)gf[ks][j] * (float)of[ks][j];
    my_delta += __shfl_xor(my_delta, 16, 64);
    my_delta += __shfl_xor(my_delta, 32, 64);
    if (g == 0 && q < s.L) delta[((int64_t)b * s.h + h) * s.L + q] = my_delta;
    my_lse2 *= LOG2E;
    const float scale2 = s.scale * LOG2E;

    // per-lane visibility limit: key visible  <=>  key < klim
    int klim = kend;
    if (s.ilk_diag >= 0) klim = min(klim, q + s.ilk_diag + 1);
    if (q >= s.L) klim = 0;
    // every lane of the wave sees all keys below klim_wave (0: the wave holds rows past L -> always masked path)
    int klim_wave = kend;
    if (s.ilk_diag >= 0) klim_wave = min(klim_wave, q0 + s.ilk_diag + 1);
    if (q0 + 16 > s.L) klim_wave = 0;

    f32x4 dq[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) dq[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int sw_row = R::swz(r16);
    const int tr_row = 4 * g + (r16 >> 2);
    const int sw_tr = R::swz(tr_row);

#pragma unroll 1
    for (int tile = 0; tile < ntiles; ++tile) {
        // tile `tile` has landed once at most the DMAs of the (up to two) younger tiles are outstanding
        xa_wait_vm(ppt * min(2, ntiles - 1 - tile));
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // slot (tile+3) % 4 held tile-1, which every wave finished reading before the barrier above
        if (tile + 3 < ntiles)
            ring_stage<HD>(smem, Kb, s.k_sr, Vb, s.v_sr, bias_row, bias_len, s.Kn, tile + 3, wid, lane);
        const uint8_t* k_lds = smem + (tile % R::NST) * R::STAGE;
        const uint8_t* v_lds = k_lds + R::TILE;
        const float* b_lds = (const float*)(k_lds + 2 * R::TILE);
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            const int kb0 = tile * KT2 + sub * 32;
            if (kb0 >= kloop) break;                  // wave-uniform
            bf16x8 ka[2][KS], va[2][KS];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int rowoff = (sub * 32 + 16 * t + r16) * ROWB;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const int off = rowoff + (((4 * ks + g) ^ sw_row) << 4);
                    ka[t][ks] = lds_row8(k_lds + off);
                    va[t][ks] = lds_row8(v_lds + off);
                }
            }
            f32x4 bb[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) bb[t] = *(const f32x4*)(b_lds + sub * 32 + 16 * t + 4 * g);
            f32x4 sacc[2], pacc[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                sacc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
                pacc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    sacc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ka[t][ks], qf[ks], sacc[t], 0, 0, 0);
                    pacc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va[t][ks], gf[ks], pacc[t], 0, 0, 0);
                }
            // transposed K fragments: in flight during the softmax arithmetic
            const uint32_t trp = lds_addr(k_lds + (sub * 32 + tr_row) * ROWB + 8 * (lane & 1));
            bf16x8 kt[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                kt[nt] = lds_tr8_async<16 * ROWB>(trp + (((2 * nt + ((lane & 3) >> 1)) ^ sw_tr) << 4));
            // dS^T = P^T * (dP^T - delta), P^T = exp2(S^T*scale2 + bias2 - lse2); masks only where a step needs them
            bf16x8 df;
            const bool plain = (kb0 + 32 <= klim_wave) && (!HAS_BIAS || kb0 + 32 <= bias_len);    // wave-uniform
            if (plain) {
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float c = HAS_BIAS ? __builtin_fmaf(bb[t][j], LOG2E, -my_lse2) : -my_lse2;
                        const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[t][j], scale2, c));
                        if constexpr (DROP)
                            df[4 * t + j] = (__bf16)(p * (pacc[t][j] * drop.km(q, kb0 + 16 * t + 4 * g + j) - my_delta));
                        else df[4 * t + j] = (__bf16)(p * (pacc[t][j] - my_delta));
                    }
            } else {
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int key = kb0 + 16 * t + 4 * g + j;
                        const float bj = (HAS_BIAS && key < bias_len) ? bb[t][j] * LOG2E : 0.f;
                        float p = __builtin_amdgcn_exp2f(sacc[t][j] * scale2 + bj - my_lse2);
                        p = (key < klim) ? p : 0.f;
                        if constexpr (DROP) df[4 * t + j] = (__bf16)(p * (pacc[t][j] * drop.km(q, key) - my_delta));
                        else df[4 * t + j] = (__bf16)(p * (pacc[t][j] - my_delta));
                    }
            }
            xa_wait_lds();
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                dq[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kt[nt], df, dq[nt], 0, 0, 0);
        }
    }

    if (q < s.L) {
        __bf16* drow = dQ + (int64_t)b * s.q_sb + (int64_t)h * s.q_sh + (int64_t)q * s.q_sr + 4 * g;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            bf16x4 o4;
#pragma unroll
            for (int j = 0; j < 4; ++j) o4[j] = (__bf16)(dq[nt][j] * s.scale);
            *(bf16x4*)(drow + nt * 16) = o4;
        }
    }
}

// ============================================================================================ dK, dV, dbias
constexpr int KB = 128;   // keys per workgroup (32 per wave)
constexpr int QS = 32;    // query rows per slice

// General variant (any L, HD in {64, 96, 128}): one workgroup = 4 waves = 128 keys; every wave keeps the K (and,
// for HD <= 96, V) MFMA fragments of its 32 keys in VGPRs, loaded straight from global memory, with dK^T, dV^T in
// its accumulators, and walks all query slices of a query block staged in LDS (which the transposed reads need).
// Two to three workgroups share a CU so that their load / compute / store phases overlap.
template <int HD>
struct DkvRegLds {
    static constexpr int PT = Pitch<HD>::TR;
    static constexpr int QB = (HD > 96) ? 64 : 128;              // query rows resident in LDS
    static constexpr bool V_LDS = HD > 96;                       // HD = 128: V stays in LDS (register budget)
    static constexpr int V_BYTES = V_LDS ? KB * Pitch<HD>::ROW : 0;
    static constexpr int BYTES = 2 * QB * PT + 2 * QB * 4 + V_BYTES;
};

template <int HD, bool DROP>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void xattn_bwd_dkv_reg_kernel(
    const __bf16* __restrict__ Q, const __bf16* __restrict__ K, const __bf16* __restrict__ V,
    const __bf16* __restrict__ dO, const float* __restrict__ lse, const float* __restrict__ delta,
    const float* __restrict__ bias, const int32_t* __restrict__ key_len, star_xattn_shape s, __bf16* __restrict__ dK,
    __bf16* __restrict__ dV, float* __restrict__ dbias_part) {
    constexpr int NTH = 256, QBLK = DkvRegLds<HD>::QB;
    constexpr int PT = Pitch<HD>::TR;
    constexpr int KS = HD / 16, NT = HD / 32;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t* q_lds = smem;                       // [QBLK][PT]
    uint8_t* g_lds = q_lds + QBLK * PT;          // [QBLK][PT]
    float* lse_lds = (float*)(g_lds + QBLK * PT);  // [QBLK]  (log2 domain)
    float* del_lds = lse_lds + QBLK;             // [QBLK]
    uint8_t* v_lds = (uint8_t*)(del_lds + QBLK); // [KB][PR] (only if V_LDS)
    constexpr bool V_LDS = DkvRegLds<HD>::V_LDS;
    constexpr int PR = Pitch<HD>::ROW;

    const int tid = threadIdx.x, lane = tid & 63, kw = tid >> 6;
    const int r = lane & 31, hf = lane >> 5;
    const int b = blockIdx.z, h = blockIdx.y;
    const int kb0 = blockIdx.x * KB;
    const int key = kb0 + kw * 32 + r;           // this lane's key

    const __bf16* Qb = Q + (int64_t)b * s.q_sb + (int64_t)h * s.q_sh;
    const __bf16* dOb = dO + (int64_t)b * s.o_sb + (int64_t)h * s.o_sh;

    int kend = s.Kn;
    if (key_len) kend = min(kend, key_len[b]);
    const bool key_ok = key < kend;
    const bool wg_live = kb0 < kend;
    const float scale2 = s.scale * LOG2E;
    const Drop<DROP> drop(s, b, h);

    // K / V fragments of this lane's key (rows past Kn are clamped: their probabilities are masked to 0)
    bf16x8 kfr[KS], vfr[V_LDS ? 1 : KS];
    {
        const int kk = min(key, s.Kn - 1);
        const __bf16* krow = K + (int64_t)b * s.k_sb + (int64_t)h * s.k_sh + (int64_t)kk * s.k_sr;
        const __bf16* vrow = V + (int64_t)b * s.v_sb + (int64_t)h * s.v_sh + (int64_t)kk * s.v_sr;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            kfr[ks] = *(const bf16x8*)(krow + (ks * 2 + hf) * 8);
            if constexpr (!V_LDS) vfr[ks] = *(const bf16x8*)(vrow + (ks * 2 + hf) * 8);
        }
        if constexpr (V_LDS) {
            if (wg_live) {
                TileRegs<HD, KB, NTH> vreg;
                tile_load<HD, KB, NTH>(vreg, V + (int64_t)b * s.v_sb + (int64_t)h * s.v_sh, s.v_sr, kb0, s.Kn);
                tile_store<HD, KB, NTH>(vreg, v_lds, PR);
            }
        }
    }
    float my_bias2 = 0.f;
    if (bias && key < s.bias_len) my_bias2 = bias[(int64_t)b * s.bias_len + key] * LOG2E;

    f32x16 dk[NT], dv[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int i = 0; i < 16; ++i) { dk[nt][i] = 0.f; dv[nt][i] = 0.f; }
    float db = 0.f;

    // with the ILK mask key j is seen only by queries i >= j - diag: skip the query blocks before that
    int qb_begin = 0;
    if (s.ilk_diag >= 0) qb_begin = max(0, (kb0 - s.ilk_diag) / QBLK * QBLK);

    for (int qb0 = qb_begin; wg_live && qb0 < s.L; qb0 += QBLK) {
        if (qb0 != qb_begin) __syncthreads();    // previous block fully consumed
        {
            TileRegs<HD, QBLK, NTH> qreg, greg;
            tile_load<HD, QBLK, NTH>(qreg, Qb, s.q_sr, qb0, s.L);
            tile_load<HD, QBLK, NTH>(greg, dOb, s.o_sr, qb0, s.L);
            float lreg = 0.f, dreg = 0.f;
            if (tid < QBLK) {
                const int qq = min(qb0 + tid, s.L - 1);
                lreg = lse[((int64_t)b * s.h + h) * s.L + qq] * LOG2E;
                dreg = delta[((int64_t)b * s.h + h) * s.L + qq];
            }
            tile_store<HD, QBLK, NTH>(qreg, q_lds, PT);
            tile_store<HD, QBLK, NTH>(greg, g_lds, PT);
            if (tid < QBLK) {
                lse_lds[tid] = lreg;
                del_lds[tid] = dreg;
            }
        }
        __syncthreads();
        const int nslices = (min(QBLK, s.L - qb0) + QS - 1) / QS;
        for (int sl = 0; sl < nslices; ++sl) {
            const int q0 = qb0 + sl * QS;
            if (s.ilk_diag >= 0 && kb0 + kw * 32 > q0 + QS - 1 + s.ilk_diag) continue;   // wave sees nothing here
            const uint8_t* qs_lds = q_lds + sl * QS * PT;
            const uint8_t* gs_lds = g_lds + sl * QS * PT;
            f32x16 sacc, pacc;
#pragma unroll
            for (int i = 0; i < 16; ++i) { sacc[i] = 0.f; pacc[i] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                bf16x8 qa = lds_row8(qs_lds + r * PT + (ks * 2 + hf) * 16);
                sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa, kfr[ks], sacc, 0, 0, 0);
                bf16x8 ga = lds_row8(gs_lds + r * PT + (ks * 2 + hf) * 16);
                bf16x8 vf;
                if constexpr (V_LDS) vf = lds_row8(v_lds + (kw * 32 + r) * PR + (ks * 2 + hf) * 16);
                else vf = vfr[ks];
                pacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ga, vf, pacc, 0, 0, 0);
            }
            const bool need_mask = !key_ok || (q0 + QS > s.L) || (s.ilk_diag >= 0 && kb0 + kw * 32 + 31 > q0 + s.ilk_diag);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 ll = *(const f32x4*)(lse_lds + sl * QS + 8 * g + 4 * hf);
                const f32x4 dd = *(const f32x4*)(del_lds + sl * QS + 8 * g + 4 * hf);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int i = g * 4 + j;
                    float p = __builtin_amdgcn_exp2f(sacc[i] * scale2 + my_bias2 - ll[j]);
                    if (need_mask) {
                        const int qq = q0 + crow(i, hf);
                        const bool vis = key_ok && (qq < s.L) && (s.ilk_diag < 0 || key <= qq + s.ilk_diag);
                        p = vis ? p : 0.f;
                    }
                    float ds;
                    if constexpr (DROP) {
                        const float km = drop.km(q0 + crow(i, hf), key);
                        ds = p * (pacc[i] * km - dd[j]);
                        p *= km;                                    // dV sees the dropped, rescaled weights
                    } else {
                        ds = p * (pacc[i] - dd[j]);
                    }
                    sacc[i] = p;
                    pacc[i] = ds;
                    db += ds;
                }
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const bf16x8 pf = pack8(sacc, s2);
                const bf16x8 df = pack8(pacc, s2);
                const int qrow = s2 * 16 + 4 * hf + ((lane & 15) >> 2);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const int col = nt * 32 + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
                    const uint8_t* g0 = gs_lds + qrow * PT + col * 2;
                    bf16x8 gt = lds_tr8(g0, g0 + 8 * PT);
                    dv[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gt, pf, dv[nt], 0, 0, 0);
                    const uint8_t* q0p = qs_lds + qrow * PT + col * 2;
                    bf16x8 qt = lds_tr8(q0p, q0p + 8 * PT);
                    dk[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qt, df, dk[nt], 0, 0, 0);
                }
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

// Short-target variant (L <= 128, HD = 64 or 128): the whole Q / dO / LSE / delta block of one (batch, head) is
// staged ONCE into LDS (XOR-swizzled, unpadded) and then every wave walks its own 16-key groups on its own,
// without a single barrier: K / V fragments of the NEXT group are prefetched into registers while the current
// group is being computed (16x16x32 MFMAs, keys on the MFMA column), and the dK / dV stores of the previous
// group drain behind the next group's arithmetic.  Load, compute and store phases therefore overlap inside a
// wave instead of relying on co-resident workgroups.  QR = 256 is the same kernel for targets of up to 256 tokens
// (133 KB of LDS at HD = 128): configs[3]'s L = 227 then avoids the general dK/dV kernel (29 -> see DESIGN.md).
template <int HD, int QR = 128>
struct DkvSmall {
    static constexpr int QROWS = QR;
    static constexpr int ROWB = HD * 2;
    static constexpr int BYTES = 2 * QROWS * ROWB + 2 * QROWS * 4;
    static constexpr int WAVES = 8;
    // 16-byte chunk swizzle of row r: conflict-free for the b128 row reads (lane groups of the 16x16x32 A operand)
    // and for the transposed b64 reads (8 rows x 32 B per half wave).  Bit 0 (HD = 128): without it the 16-byte K park
    // stores of the fused backward -- rows r and r + 4 of an 8-lane group -- were 2-way on the 128-byte store window;
    // the reads are conflict-free with or without it
    __device__ static __forceinline__ int swz(int r) {
        if constexpr (HD == 128) return ((r & 7) << 1) | ((r >> 2) & 1);
        else return (((r >> 1) & 1) << 2) | (((r >> 2) & 1) << 1) | ((r >> 2) & 1);
    }
};

template <int HD, bool DROP, int QR = 128>
__global__ __launch_bounds__(512) void xattn_bwd_dkv_small_kernel(
    const __bf16* __restrict__ Q, const __bf16* __restrict__ K, const __bf16* __restrict__ V,
    const __bf16* __restrict__ dO, const float* __restrict__ lse, const float* __restrict__ delta,
    const float* __restrict__ bias, const int32_t* __restrict__ key_len, star_xattn_shape s, __bf16* __restrict__ dK,
    __bf16* __restrict__ dV, float* __restrict__ dbias_part) {
    using C = DkvSmall<HD, QR>;
    constexpr int NTH = 512, QROWS = C::QROWS, ROWB = C::ROWB;
    constexpr int KS = HD / 32;       // k-steps of a 16x16x32 MFMA over the head dim
    constexpr int NT = HD / 16;       // 16-wide head-dim tiles of dK^T / dV^T
    constexpr int CPR = HD / 8;       // 16-byte chunks per row
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t* q_lds = smem;                          // [128][ROWB] swizzled
    uint8_t* g_lds = q_lds + QROWS * ROWB;          // [128][ROWB] swizzled
    float* lse_lds = (float*)(g_lds + QROWS * ROWB);  // [128] (log2 domain)
    float* del_lds = lse_lds + QROWS;               // [128]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);     // wave-uniform: group loops and mask tests on SALU
    const int r16 = lane & 15, g = lane >> 4;
    const int b = blockIdx.z, h = blockIdx.y;

    const __bf16* Qb = Q + (int64_t)b * s.q_sb + (int64_t)h * s.q_sh;
    const __bf16* dOb = dO + (int64_t)b * s.o_sb + (int64_t)h * s.o_sh;
    const __bf16* Kb = K + (int64_t)b * s.k_sb + (int64_t)h * s.k_sh;
    const __bf16* Vb = V + (int64_t)b * s.v_sb + (int64_t)h * s.v_sh;
    __bf16* dKb = dK + (int64_t)b * s.k_sb + (int64_t)h * s.k_sh;
    __bf16* dVb = dV + (int64_t)b * s.v_sb + (int64_t)h * s.v_sh;

    int kend = s.Kn;
    if (key_len) kend = min(kend, key_len[b]);
    const float scale2 = s.scale * LOG2E;
    const Drop<DROP> drop(s, b, h);

    const int ngroups = (s.Kn + 15) >> 4;
    const int gstride = gridDim.x * C::WAVES;
    int grp = blockIdx.x * C::WAVES + wid;

    // first group's K / V fragments: issued before the Q / dO staging so that both are in flight together
    bf16x8 kfr[KS], vfr[KS];
    float bias_cur = 0.f;
    {
        const int key = min(grp * 16 + r16, s.Kn - 1);
        const __bf16* krow = Kb + (int64_t)key * s.k_sr + g * 8;
        const __bf16* vrow = Vb + (int64_t)key * s.v_sr + g * 8;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            kfr[ks] = *(const bf16x8*)(krow + ks * 32);
            vfr[ks] = *(const bf16x8*)(vrow + ks * 32);
        }
        if (bias && key < s.bias_len) bias_cur = bias[(int64_t)b * s.bias_len + key];
    }
    {
        TileRegs<HD, QROWS, NTH> qreg, greg;
        tile_load<HD, QROWS, NTH>(qreg, Qb, s.q_sr, 0, s.L);
        tile_load<HD, QROWS, NTH>(greg, dOb, s.o_sr, 0, s.L);
        float lreg = 0.f, dreg = 0.f;
        if (tid < QROWS) {
            const int qq = min(tid, s.L - 1);
            lreg = lse[((int64_t)b * s.h + h) * s.L + qq] * LOG2E;
            dreg = delta[((int64_t)b * s.h + h) * s.L + qq];
        }
#pragma unroll
        for (int i = 0; i < TileRegs<HD, QROWS, NTH>::N; ++i) {
            const int c = tid + i * NTH;
            const int row = c / CPR, ch = c % CPR;
            const int off = row * ROWB + ((ch ^ C::swz(row & 15)) << 4);
            *(u32x4*)(q_lds + off) = qreg.v[i];
            *(u32x4*)(g_lds + off) = greg.v[i];
        }
        if (tid < QROWS) {
            lse_lds[tid] = lreg;
            del_lds[tid] = dreg;
        }
    }
    __syncthreads();          // the only barrier: from here on the waves run independently

    const int nslices = (s.L + 31) >> 5;
    // per-lane LDS offsets that do not depend on the group
    const int sw_row = C::swz(r16);                               // row reads: row = 16t + r16
    const int tr_row = 4 * g + (r16 >> 2);                        // transposed reads: row within the slice
    const int sw_tr = C::swz(tr_row);                             // (row & 15) is the same for row and row + 16

    for (; grp < ngroups; grp += gstride) {
        const int key = grp * 16 + r16;
        const bool key_ok = key < kend;
        const float my_bias2 = bias_cur * LOG2E;
        // smallest query that sees this lane's key (INT_MAX: key masked out altogether)
        const int qmin = key_ok ? (s.ilk_diag >= 0 ? key - s.ilk_diag : -(1 << 30)) : 0x7fffffff;
        // prefetch the next group's fragments
        bf16x8 kn[KS], vn[KS];
        float bias_nxt = 0.f;
        {
            const int ng = grp + gstride;
            const int nkey = min((ng < ngroups ? ng : grp) * 16 + r16, s.Kn - 1);
            const __bf16* krow = Kb + (int64_t)nkey * s.k_sr + g * 8;
            const __bf16* vrow = Vb + (int64_t)nkey * s.v_sr + g * 8;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                kn[ks] = *(const bf16x8*)(krow + ks * 32);
                vn[ks] = *(const bf16x8*)(vrow + ks * 32);
            }
            if (bias && nkey < s.bias_len) bias_nxt = bias[(int64_t)b * s.bias_len + nkey];
        }

        f32x4 dk[NT], dv[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) { dk[nt] = (f32x4){0.f, 0.f, 0.f, 0.f}; dv[nt] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
        float db = 0.f;

        if (grp * 16 < kend) {
#pragma unroll 1
            for (int sl = 0; sl < nslices; ++sl) {
                const int q0 = sl * 32;
                if (s.ilk_diag >= 0 && grp * 16 > q0 + 31 + s.ilk_diag) continue;      // nothing visible (uniform)
                // S[q][key], dP[q][key] for two 16-query tiles (key on the lane, 4 queries per tile in registers):
                // all sixteen A fragments are requested before the first MFMA
                // dropout build: the swizzled fragment offsets are re-formed per slice (see the fused kernel)
                int sw_tr_v = sw_tr, sw_row_v = sw_row;
                if constexpr (DROP) asm volatile("" : "+v"(sw_tr_v), "+v"(sw_row_v));
                bf16x8 qa[2][KS], ga[2][KS];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int rowoff = (q0 + 16 * t + r16) * ROWB;
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) {
                        const int off = rowoff + (((4 * ks + g) ^ sw_row_v) << 4);
                        qa[t][ks] = lds_row8(q_lds + off);
                        ga[t][ks] = lds_row8(g_lds + off);
                    }
                }
                f32x4 ll[2], dd[2];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    ll[t] = *(const f32x4*)(lse_lds + q0 + 16 * t + 4 * g);
                    dd[t] = *(const f32x4*)(del_lds + q0 + 16 * t + 4 * g);
                }
                f32x4 sacc[2], pacc[2];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    sacc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    pacc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        sacc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa[t][ks], kfr[ks], sacc[t], 0, 0, 0);
                        pacc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ga[t][ks], vfr[ks], pacc[t], 0, 0, 0);
                    }
                // first half of the transposed fragments: in flight during the softmax arithmetic
                const int trbase = (q0 + tr_row) * ROWB + 8 * (lane & 1);
                bf16x8 gt[NT / 2], qt[NT / 2];
#pragma unroll
                for (int nt = 0; nt < NT / 2; ++nt) {
                    const int off = trbase + (((2 * nt + ((lane & 3) >> 1)) ^ sw_tr_v) << 4);
                    gt[nt] = lds_tr8(g_lds + off, g_lds + off + 16 * ROWB);
                    qt[nt] = lds_tr8(q_lds + off, q_lds + off + 16 * ROWB);
                }
                // visibility of (query, key): key valid, query < L, key <= query + diag  <=>  qmin <= query < L
                bf16x8 pf, df;
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int qq = q0 + 16 * t + 4 * g + j;
                        float p = __builtin_amdgcn_exp2f(sacc[t][j] * scale2 + my_bias2 - ll[t][j]);
                        p = (qq >= qmin && qq < s.L) ? p : 0.f;
                        float ds;
                        if constexpr (DROP) {
                            const float km = drop.km(qq, key);
                            ds = p * (pacc[t][j] * km - dd[t][j]);
                            p *= km;
                        } else {
                            ds = p * (pacc[t][j] - dd[t][j]);
                        }
                        db += ds;
                        pf[4 * t + j] = (__bf16)p;
                        df[4 * t + j] = (__bf16)ds;
                    }
                // dV^T[n][key] += sum_q dO[q][n] P[q][key];  dK^T[n][key] += sum_q Q[q][n] dS[q][key]
#pragma unroll
                for (int hlf = 0; hlf < 2; ++hlf) {
                    bf16x8 gt2[NT / 2], qt2[NT / 2];
                    if (hlf == 0) {
#pragma unroll
                        for (int nt = 0; nt < NT / 2; ++nt) {
                            const int off = trbase + (((2 * (nt + NT / 2) + ((lane & 3) >> 1)) ^ sw_tr_v) << 4);
                            gt2[nt] = lds_tr8(g_lds + off, g_lds + off + 16 * ROWB);
                            qt2[nt] = lds_tr8(q_lds + off, q_lds + off + 16 * ROWB);
                        }
                    }
#pragma unroll
                    for (int nt = 0; nt < NT / 2; ++nt) {
                        dv[nt + hlf * (NT / 2)] =
                            __builtin_amdgcn_mfma_f32_16x16x32_bf16(gt[nt], pf, dv[nt + hlf * (NT / 2)], 0, 0, 0);
                        dk[nt + hlf * (NT / 2)] =
                            __builtin_amdgcn_mfma_f32_16x16x32_bf16(qt[nt], df, dk[nt + hlf * (NT / 2)], 0, 0, 0);
                    }
                    if (hlf == 0) {
#pragma unroll
                        for (int nt = 0; nt < NT / 2; ++nt) { gt[nt] = gt2[nt]; qt[nt] = qt2[nt]; }
                    }
                }
            }
        }

        // Take over the prefetched fragments BEFORE this group's stores are issued: vmcnt retires in order, so a wait
        // for these loads placed after the stores (where the compiler puts it on its own) would also wait for
        // the 33 stores to complete, once per group.
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            kfr[ks] = kn[ks];
            vfr[ks] = vn[ks];
            asm volatile("" : "+v"(kfr[ks]), "+v"(vfr[ks]));
        }
        bias_cur = bias_nxt;
        asm volatile("" : "+v"(bias_cur));

        if (key < s.Kn) {
            __bf16* dkrow = dKb + (int64_t)key * s.k_sr + 4 * g;
            __bf16* dvrow = dVb + (int64_t)key * s.v_sr + 4 * g;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                bf16x4 a, c;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    a[j] = (__bf16)(dk[nt][j] * s.scale);
                    c[j] = (__bf16)dv[nt][j];
                }
                *(bf16x4*)(dkrow + nt * 16) = a;
                *(bf16x4*)(dvrow + nt * 16) = c;
            }
        }
        db += __shfl_xor(db, 16, 64);
        db += __shfl_xor(db, 32, 64);
        if (dbias_part && g == 0 && key < s.Kn) dbias_part[((int64_t)b * s.h + h) * s.Kn + key] = db;
    }
}

// Fused short-target backward (L <= 128, HD = 64 or 128; all but tiny batches with very long memories): dQ, dK, dV
// and the bias-gradient partials of one (batch, head) from ONE pass over K / V.  The two products want opposite
// ownerships (dK/dV: a wave owns keys and sees every query; dQ: a wave owns queries and sees every key), so the keys
// are walked in rounds of 128 = one 16-key group per wave:
//   phase B  the dkv_small walk for the wave's group (S, dP, dS, dK^T, dV^T); on the way the wave parks its K
//            fragment and its bf16 dS block [16 keys][128 queries] in LDS
//   barrier
//   phase C  wave w now owns queries 16w..16w+15: dQ^T[hd][q] += K^T[hd][key] dS^T[key][q] over the round's keys,
//            both operands by transposed LDS reads (the same key permutation on both sides)
//   dK / dV stores of the round, barrier (LDS is re-used by the next round)
// delta = rowsum(dO * O) is formed while the Q / dO block is staged.  Compared with the dQ + dK/dV pair this reads
// K, V, Q, dO once instead of twice and recomputes S / dP once instead of twice.
template <int HD>
struct BwdFused {
    static constexpr int QROWS = 128, KROWS = 128, WAVES = 8;
    static constexpr int ROWB = HD * 2;
    static constexpr int DSB = QROWS * 2;                       // one dS row: 128 queries of one key
    static constexpr int K_OFF = 2 * QROWS * ROWB;
    static constexpr int DS_OFF = K_OFF + KROWS * ROWB;
    static constexpr int LSE_OFF = DS_OFF + KROWS * DSB;
    static constexpr int BYTES = LSE_OFF + 2 * QROWS * 4;
};

__device__ __forceinline__ void xa_lds_barrier() {
    // LDS traffic of this wave done, then the workgroup barrier; global stores / prefetches stay in flight
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <int HD, bool DROP>
__global__ __launch_bounds__(512) void xattn_bwd_fused_small_kernel(
    const __bf16* __restrict__ Q, const __bf16* __restrict__ K, const __bf16* __restrict__ V,
    const __bf16* __restrict__ O, const __bf16* __restrict__ dO, const float* __restrict__ lse,
    const float* __restrict__ bias, const int32_t* __restrict__ key_len, star_xattn_shape s, __bf16* __restrict__ dQ,
    __bf16* __restrict__ dK, __bf16* __restrict__ dV, float* __restrict__ dbias_part) {
    using C = DkvSmall<HD>;
    using F = BwdFused<HD>;
    constexpr int NTH = 512, QROWS = F::QROWS, ROWB = F::ROWB, DSB = F::DSB;
    constexpr int KS = HD / 32;       // k-steps of a 16x16x32 MFMA over the head dim
    constexpr int NT = HD / 16;       // 16-wide head-dim tiles of dK^T / dV^T / dQ^T
    constexpr int CPR = HD / 8;       // 16-byte chunks per row
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t* q_lds = smem;                          // [128 queries][ROWB] swizzled
    uint8_t* g_lds = q_lds + QROWS * ROWB;          // [128 queries][ROWB] swizzled
    uint8_t* k_lds = smem + F::K_OFF;               // [128 keys of the round][ROWB] swizzled
    uint8_t* ds_lds = smem + F::DS_OFF;             // [128 keys of the round][128 queries] bf16, swizzled
    float* lse_lds = (float*)(smem + F::LSE_OFF);   // [128] (log2 domain)
    float* del_lds = lse_lds + QROWS;               // [128]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, g = lane >> 4;
    const int b = blockIdx.z, h = blockIdx.y;

    const __bf16* Qb = Q + (int64_t)b * s.q_sb + (int64_t)h * s.q_sh;
    const __bf16* Ob = O + (int64_t)b * s.o_sb + (int64_t)h * s.o_sh;
    const __bf16* dOb = dO + (int64_t)b * s.o_sb + (int64_t)h * s.o_sh;
    const __bf16* Kb = K + (int64_t)b * s.k_sb + (int64_t)h * s.k_sh;
    const __bf16* Vb = V + (int64_t)b * s.v_sb + (int64_t)h * s.v_sh;
    __bf16* dKb = dK + (int64_t)b * s.k_sb + (int64_t)h * s.k_sh;
    __bf16* dVb = dV + (int64_t)b * s.v_sb + (int64_t)h * s.v_sh;

    int kend = s.Kn;
    if (key_len) kend = min(kend, key_len[b]);
    const float scale2 = s.scale * LOG2E;
    const Drop<DROP> drop(s, b, h);
    const int ngroups = (s.Kn + 15) >> 4;
    const int nrounds = (s.Kn + F::KROWS - 1) / F::KROWS;

    XA_STAMP(0);
    // first group's K / V fragments: issued before the Q / dO staging so that both are in flight together
    bf16x8 kfr[KS], vfr[KS];
    float bias_cur = 0.f;
    {
        const int key = min(wid * 16 + r16, s.Kn - 1);
        const __bf16* krow = Kb + (int64_t)key * s.k_sr + g * 8;
        const __bf16* vrow = Vb + (int64_t)key * s.v_sr + g * 8;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            kfr[ks] = *(const bf16x8*)(krow + ks * 32);
            vfr[ks] = *(const bf16x8*)(vrow + ks * 32);
        }
        if (bias && key < s.bias_len) bias_cur = bias[(int64_t)b * s.bias_len + key];
    }
    {
        TileRegs<HD, QROWS, NTH> qreg, greg, oreg;
        tile_load<HD, QROWS, NTH>(qreg, Qb, s.q_sr, 0, s.L);
        tile_load<HD, QROWS, NTH>(greg, dOb, s.o_sr, 0, s.L);
        tile_load<HD, QROWS, NTH>(oreg, Ob, s.o_sr, 0, s.L);
        float lreg = 0.f;
        if (tid < QROWS) lreg = lse[((int64_t)b * s.h + h) * s.L + min(tid, s.L - 1)] * LOG2E;
#pragma unroll
        for (int i = 0; i < TileRegs<HD, QROWS, NTH>::N; ++i) {
            const int c = tid + i * NTH;
            const int row = c / CPR, ch = c % CPR;
            const int off = row * ROWB + ((ch ^ C::swz(row & 15)) << 4);
            *(u32x4*)(q_lds + off) = qreg.v[i];
            *(u32x4*)(g_lds + off) = greg.v[i];
            // delta[row] = sum over the row of dO * O: the CPR lanes that hold one row are neighbours
            const bf16x8 gv = __builtin_bit_cast(bf16x8, greg.v[i]);
            const bf16x8 ov = __builtin_bit_cast(bf16x8, oreg.v[i]);
            float d = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) d += (float)gv[j] * (float)ov[j];
#pragma unroll
            for (int m = 1; m < CPR; m <<= 1) d += __shfl_xor(d, m, 64);
            if (ch == 0) del_lds[row] = d;
        }
        if (tid < QROWS) lse_lds[tid] = lreg;
    }
    __syncthreads();
    XA_STAMP(1);

    const int nslices = (s.L + 31) >> 5;
    const int sw_row = C::swz(r16);                               // row reads: row = 16t + r16
    const int tr_row = 4 * g + (r16 >> 2);                        // transposed reads: row within a 32-row slice
    const int sw_tr = C::swz(tr_row);
    // dS tile (256-byte rows, 8-byte units = 4 queries of one key): chunk ^ (((key & 7) << 1) | ((key >> 2) & 1)), half
    // of the chunk ^ ((key >> 3) & 1).  The 16 keys that one ds_write_b64 lane group stores then land on 16 distinct
    // 8-byte slots of the 128-byte bank window (with the chunk swizzle alone: 4-way, 96 extra LDS cycles per wave and
    // round), and the transposed reads of phase C still find each key's 32 bytes in one piece (conflict-free as before)
    const int sw_ds_row = ((r16 & 7) << 1) | ((r16 >> 2) & 1);
    const int hf_ds_row = (r16 >> 3) & 1;
    const int sw_ds_tr = ((tr_row & 7) << 1) | ((tr_row >> 2) & 1);
    const int hf_ds_tr = (tr_row >> 3) & 1;
    const int qw0 = wid * 16;                                     // phase C: first query of this wave
    const bool wave_has_q = qw0 < s.L;

    f32x4 dq[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) dq[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

#pragma unroll 1
    for (int rnd = 0; rnd < nrounds; ++rnd) {
        const int grp = rnd * F::WAVES + wid;
        const int key = grp * 16 + r16;
        const bool key_ok = key < kend;
        const float my_bias2 = bias_cur * LOG2E;
        const int qmin = key_ok ? (s.ilk_diag >= 0 ? key - s.ilk_diag : -(1 << 30)) : 0x7fffffff;
        // park this group's K rows for phase C (rows past Kn hold a clamped, finite row: their dS is zero)
        uint8_t* krow_lds = k_lds + (16 * wid + r16) * ROWB;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) *(bf16x8*)(krow_lds + (((4 * ks + g) ^ sw_row) << 4)) = kfr[ks];
        // HD = 128 reads its K rows back from that copy, so the next round's K fragments can be requested right away
        // (into the same registers) and have the whole slice loop to arrive
        const int ng = grp + F::WAVES;
        const int nkey = min((ng < ngroups ? ng : grp) * 16 + r16, s.Kn - 1);
        if constexpr (HD == 128) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(kfr[ks]));      // the LDS copy first
            const __bf16* krow = Kb + (int64_t)nkey * s.k_sr + g * 8;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) kfr[ks] = *(const bf16x8*)(krow + ks * 32);
        }
        uint8_t* ds_row = ds_lds + (16 * wid + r16) * DSB + 8 * ((g & 1) ^ hf_ds_row);

        f32x4 dk[NT], dv[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) { dk[nt] = (f32x4){0.f, 0.f, 0.f, 0.f}; dv[nt] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
        float db = 0.f;
        const bool grp_live = grp * 16 < kend;

        // slices whose queries cannot see this group (ILK: the early ones; a group past the last key: all of them)
        // only leave zeros in the dS tile.  Two loops, not a `continue`: a skip path inside the compute loop makes
        // the compiler copy all 64 accumulator registers around the back edge of every iteration.
        int sl_first = (s.ilk_diag >= 0) ? (max(grp * 16 - s.ilk_diag, 0) >> 5) : 0;
        if (!grp_live) sl_first = nslices;
        sl_first = min(sl_first, nslices);
#pragma unroll 1
        for (int sl = 0; sl < sl_first; ++sl) {
            *(uint2*)(ds_row + (((4 * sl + (g >> 1)) ^ sw_ds_row) << 4)) = make_uint2(0u, 0u);
            *(uint2*)(ds_row + (((4 * sl + 2 + (g >> 1)) ^ sw_ds_row) << 4)) = make_uint2(0u, 0u);
        }
#pragma unroll 1
        for (int sl = sl_first; sl < nslices; ++sl) {
            const int q0 = sl * 32;
            // dropout build: the eight swizzled fragment offsets are re-formed in every slice instead of living in
            // registers across the round (with the mask arithmetic they no longer fit: 14 spilled VGPRs, whose reloads
            // wait vmcnt(0) and so drain the K / V prefetch)
            int sw_tr_v = sw_tr, sw_row_v = sw_row;
            if constexpr (DROP) asm volatile("" : "+v"(sw_tr_v), "+v"(sw_row_v));
            const int dsc0 = ((4 * sl + (g >> 1)) ^ sw_ds_row) << 4, dsc1 = ((4 * sl + 2 + (g >> 1)) ^ sw_ds_row) << 4;
            // no mask needed: every key of the group is valid and visible to every query of the slice
            const bool plain = (grp * 16 + 16 <= kend) && (q0 + 32 <= s.L) &&
                               (s.ilk_diag < 0 || grp * 16 + 15 - s.ilk_diag <= q0);
            // S[q][key], dP[q][key] and the softmax arithmetic for two 16-query tiles, one after the other (the
            // HD = 128 kernel has no room for the fragments of both at once next to the dK / dV / dQ accumulators).
            // The wave's own K rows come back from the LDS copy made for phase C (program order: no barrier needed),
            // which frees their registers for the whole slice loop.
            bf16x8 kb[KS];
            if constexpr (HD == 128) {
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) kb[ks] = lds_row8(krow_lds + (((4 * ks + g) ^ sw_row_v) << 4));
            } else {
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) kb[ks] = kfr[ks];
            }
            // transposed dO / Q fragments in four steps of TPS head-dim tiles: the first step is in flight during the
            // softmax arithmetic, every further one behind the MFMAs of the step before it
            constexpr int TPS = NT / 4;
            const int trbase = (q0 + tr_row) * ROWB + 8 * (lane & 1);
            bf16x8 gt[TPS], qt[TPS];
            bf16x8 pf, df;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                bf16x8 qa[KS], ga[KS];
                const int rowoff = (q0 + 16 * t + r16) * ROWB;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const int off = rowoff + (((4 * ks + g) ^ sw_row_v) << 4);
                    qa[ks] = lds_row8(q_lds + off);
                    ga[ks] = lds_row8(g_lds + off);
                }
                const f32x4 ll = *(const f32x4*)(lse_lds + q0 + 16 * t + 4 * g);
                const f32x4 dd = *(const f32x4*)(del_lds + q0 + 16 * t + 4 * g);
                f32x4 sacc = (f32x4){0.f, 0.f, 0.f, 0.f}, pacc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    sacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa[ks], kb[ks], sacc, 0, 0, 0);
                    pacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ga[ks], vfr[ks], pacc, 0, 0, 0);
                }
                if (t == 1) {
#pragma unroll
                    for (int nt = 0; nt < TPS; ++nt) {
                        const int off = trbase + (((2 * nt + ((lane & 3) >> 1)) ^ sw_tr_v) << 4);
                        gt[nt] = lds_tr8(g_lds + off, g_lds + off + 16 * ROWB);
                        qt[nt] = lds_tr8(q_lds + off, q_lds + off + 16 * ROWB);
                    }
                }
                if (plain) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float p = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[j], scale2, my_bias2 - ll[j]));
                        float ds;
                        if constexpr (DROP) {
                            const float km = drop.km(q0 + 16 * t + 4 * g + j, key);
                            ds = p * (pacc[j] * km - dd[j]);
                            p *= km;
                        } else {
                            ds = p * (pacc[j] - dd[j]);
                        }
                        db += ds;
                        pf[4 * t + j] = (__bf16)p;
                        df[4 * t + j] = (__bf16)ds;
                    }
                } else {
                    // visibility of (query, key): key valid, query < L, key <= query + diag  <=>  qmin <= query < L
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int qq = q0 + 16 * t + 4 * g + j;
                        float p = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[j], scale2, my_bias2 - ll[j]));
                        const bool vis = (qq >= qmin) & (qq < s.L);
                        p = vis ? p : 0.f;
                        float ds;
                        if constexpr (DROP) {
                            const float km = drop.km(qq, key);
                            ds = p * (pacc[j] * km - dd[j]);
                            p *= km;
                        } else {
                            ds = p * (pacc[j] - dd[j]);
                        }
                        db += ds;
                        pf[4 * t + j] = (__bf16)p;
                        df[4 * t + j] = (__bf16)ds;
                    }
                }
            }
            // dS of (this key, queries q0+16t+4g .. +3) for phase C
            {
                const u32x4 dfw = __builtin_bit_cast(u32x4, df);
                *(uint2*)(ds_row + dsc0) = make_uint2(dfw[0], dfw[1]);
                *(uint2*)(ds_row + dsc1) = make_uint2(dfw[2], dfw[3]);
            }
            if constexpr (HD == 128) __builtin_amdgcn_sched_barrier(0);     // keeps the later fragment reads below
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                bf16x8 gt2[TPS], qt2[TPS];
                if (st < 3) {
#pragma unroll
                    for (int nt = 0; nt < TPS; ++nt) {
                        const int off = trbase + (((2 * (nt + (st + 1) * TPS) + ((lane & 3) >> 1)) ^ sw_tr_v) << 4);
                        gt2[nt] = lds_tr8(g_lds + off, g_lds + off + 16 * ROWB);
                        qt2[nt] = lds_tr8(q_lds + off, q_lds + off + 16 * ROWB);
                    }
                }
#pragma unroll
                for (int nt = 0; nt < TPS; ++nt) {
                    dv[nt + st * TPS] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gt[nt], pf, dv[nt + st * TPS], 0, 0, 0);
                    dk[nt + st * TPS] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qt[nt], df, dk[nt + st * TPS], 0, 0, 0);
                }
                if (st < 3) {
#pragma unroll
                    for (int nt = 0; nt < TPS; ++nt) { gt[nt] = gt2[nt]; qt[nt] = qt2[nt]; }
                }
            }
        }

        // The next round's K / V fragments go straight into the (now dead) fragment registers: the HD = 128 kernel has
        // no room for a second set next to the dQ accumulators.  They are in flight across the barrier and phase C and
        // are waited for BEFORE this round's stores are issued (vmcnt retires in order).
        {
            const __bf16* krow = Kb + (int64_t)nkey * s.k_sr + g * 8;
            const __bf16* vrow = Vb + (int64_t)nkey * s.v_sr + g * 8;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                if constexpr (HD != 128) kfr[ks] = *(const bf16x8*)(krow + ks * 32);
                vfr[ks] = *(const bf16x8*)(vrow + ks * 32);
            }
            bias_cur = 0.f;
            if (bias && nkey < s.bias_len) bias_cur = bias[(int64_t)b * s.bias_len + nkey];
        }

        XA_STAMP(2 + 4 * (rnd & 3));
        xa_lds_barrier();          // K rows and dS of the whole round are in LDS
        XA_STAMP(3 + 4 * (rnd & 3));

        // phase C: this wave's 16 queries against the round's keys
        if (wave_has_q) {
#pragma unroll 1
            for (int kk = 0; kk < F::KROWS / 32; ++kk) {
                const int k0 = rnd * F::KROWS + 32 * kk;
                if (k0 >= kend) break;                                                   // wave-uniform
                if (s.ilk_diag >= 0 && k0 > qw0 + 15 + s.ilk_diag) break;                // invisible to all 16 queries
                const int trb = (32 * kk + tr_row) * ROWB + 8 * (lane & 1);
                int sw_tr_c = sw_tr;
                if constexpr (DROP) asm volatile("" : "+v"(sw_tr_c));
                bf16x8 kt[NT];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const int off = trb + (((2 * nt + ((lane & 3) >> 1)) ^ sw_tr_c) << 4);
                    kt[nt] = lds_tr8(k_lds + off, k_lds + off + 16 * ROWB);
                }
                const int dso = (32 * kk + tr_row) * DSB + 8 * ((lane & 1) ^ hf_ds_tr) +
                                (((2 * wid + ((lane & 3) >> 1)) ^ sw_ds_tr) << 4);
                const bf16x8 dst = lds_tr8(ds_lds + dso, ds_lds + dso + 16 * DSB);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    dq[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kt[nt], dst, dq[nt], 0, 0, 0);
            }
        }

#pragma unroll
        for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(kfr[ks]), "+v"(vfr[ks]));
        asm volatile("" : "+v"(bias_cur));

        db += __shfl_xor(db, 16, 64);
        db += __shfl_xor(db, 32, 64);
        if (dbias_part && g == 0 && key < s.Kn) dbias_part[((int64_t)b * s.h + h) * s.Kn + key] = db;

        XA_STAMP(4 + 4 * (rnd & 3));
        xa_lds_barrier();      // phase C reads done: the round's LDS rows are free again
        XA_STAMP(5 + 4 * (rnd & 3));

        // dK / dV of the round leave through the wave's own 16 rows of the K and dS tiles (free until the next
        // round's writes, which follow in program order): accumulator layout -> row-major, so that every store
        // instruction writes whole 16-byte-per-lane rows instead of 32-byte fragments of 16 rows
        {
            uint8_t* tk = k_lds + 16 * wid * ROWB;
            uint8_t* tv = ds_lds + 16 * wid * DSB;
            int lanev = lane;
            asm volatile("" : "+v"(lanev));        // addresses formed here, not hoisted out of the round loop
            const int r16v = lanev & 15, gv = lanev >> 4;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                bf16x4 a, c;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    a[j] = (__bf16)(dk[nt][j] * s.scale);
                    c[j] = (__bf16)dv[nt][j];
                }
                // (the two 8-byte halves of a chunk trade places in rows 8-15: the 16 rows one ds_write_b64 lane group
                // stores then fall on 16 distinct 8-byte slots instead of 8; the row reads below swap them back)
                const int choff = (((2 * nt + (gv >> 1)) ^ (r16v & (CPR - 1))) << 4) + 8 * ((gv & 1) ^ (r16v >> 3));
                *(bf16x4*)(tk + r16v * ROWB + choff) = a;
                *(bf16x4*)(tv + r16v * DSB + choff) = c;
            }
#pragma unroll
            for (int i = 0; i < CPR / 4; ++i) {
                const int row = lanev / CPR + i * (64 / CPR), c = lanev % CPR;
                const int choff = (c ^ (row & (CPR - 1))) << 4;
                u32x4 x = *(const u32x4*)(tk + row * ROWB + choff);
                u32x4 y = *(const u32x4*)(tv + row * DSB + choff);
                if (row & 8) {
                    x = (u32x4){x[2], x[3], x[0], x[1]};
                    y = (u32x4){y[2], y[3], y[0], y[1]};
                }
                const int krow = grp * 16 + row;
                if (krow < s.Kn) {
                    *(u32x4*)(dKb + (int64_t)krow * s.k_sr + c * 8) = x;
                    *(u32x4*)(dVb + (int64_t)krow * s.v_sr + c * 8) = y;
                }
            }
        }
    }

    // dQ^T accumulators -> row-major through the same private rows (every wave passed the last barrier)
    {
        uint8_t* tq = k_lds + 16 * wid * ROWB;
        int lanev = lane;
        asm volatile("" : "+v"(lanev));
        const int r16v = lanev & 15, gv = lanev >> 4;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            bf16x4 o4;
#pragma unroll
            for (int j = 0; j < 4; ++j) o4[j] = (__bf16)(dq[nt][j] * s.scale);
            *(bf16x4*)(tq + r16v * ROWB + (((2 * nt + (gv >> 1)) ^ (r16v & (CPR - 1))) << 4) +
                       8 * ((gv & 1) ^ (r16v >> 3))) = o4;
        }
#pragma unroll
        for (int i = 0; i < CPR / 4; ++i) {
            const int row = lanev / CPR + i * (64 / CPR), c = lanev % CPR;
            u32x4 x = *(const u32x4*)(tq + row * ROWB + ((c ^ (row & (CPR - 1))) << 4));
            if (row & 8) x = (u32x4){x[2], x[3], x[0], x[1]};
            const int q = qw0 + row;
            if (q < s.L) *(u32x4*)(dQ + (int64_t)b * s.q_sb + (int64_t)h * s.q_sh + (int64_t)q * s.q_sr + c * 8) = x;
        }
    }
#ifdef STAR_XA_STAMP
    XA_STAMP(14);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    XA_STAMP(15);
#endif
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

// dbias[b,j] = sum_l sum_h part_l[b,h,j]: the bias-gradient partials of ALL decoder layers in one launch (the
// reference feeds the same scorer_logits to every layer, my_transformer.py:917-928), fixed summation order
constexpr int DBIAS_MAX_LAYERS = 32;
struct DbiasParts {
    const float* p[DBIAS_MAX_LAYERS];
};
__global__ __launch_bounds__(256) void xattn_dbias_sum_kernel(DbiasParts parts, int nl, int B, int H, int Kn,
                                                              int bias_len, float* __restrict__ dbias) {
    // 64 columns per workgroup; the nl*H (layer, head) rows are dealt round-robin to the 4 waves so that the loads
    // of a column are independent and in flight together, then the 4 partial sums are added in a fixed order
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);     // wave-uniform: pointer table read by SALU
    const int j = blockIdx.x * 64 + lane;
    const int b = blockIdx.y;
    const int jc = min(j, bias_len - 1);
    const int total = nl * H;
    float sum = 0.f;
    int n = w;
    for (; n + 12 < total; n += 16) {
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int m = n + 4 * u;
            v[u] = parts.p[m / H][((int64_t)b * H + m % H) * Kn + jc];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) sum += v[u];
    }
    for (; n < total; n += 4) sum += parts.p[n / H][((int64_t)b * H + n % H) * Kn + jc];
    red[w][lane] = sum;
    __syncthreads();
    if (w == 0 && j < bias_len) dbias[(int64_t)b * bias_len + j] = ((red[0][lane] + red[1][lane]) + red[2][lane]) + red[3][lane];
}

// P[b,h,q,key] = exp(logit - lse): the optional attention-weights output (output_attentions)
template <int HD, bool DROP>
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
    const Drop<DROP> drop(s, b, h);
    for (int key = lane; key < s.Kn; key += 64) {
        const __bf16* krow = K + (int64_t)b * s.k_sb + (int64_t)h * s.k_sh + (int64_t)key * s.k_sr;
        float acc = 0.f;
        for (int n = 0; n < HD; ++n) acc += (float)qrow[n] * (float)krow[n];
        const bool vis = (key < kend) && (s.ilk_diag < 0 || key <= q + s.ilk_diag);
        const float bj = (bias && key < s.bias_len) ? bias[(int64_t)b * s.bias_len + key] : 0.f;
        P[row * s.Kn + key] = vis ? __expf(acc * s.scale + bj - l) * drop.km(q, key) : 0.f;
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
    STAR_REQUIRE(s->dropout_p >= 0.f && s->dropout_p < 1.f, "%s: dropout_p %g not in [0, 1)", who, (double)s->dropout_p);
    STAR_REQUIRE((int64_t)s->L * s->Kn < (1ll << 32), "%s: L*Kn exceeds the 32-bit dropout counter", who);
    return STAR_OK;
}

#define XA_RESERVE(fn, bytes, who)                                                          \
    do {                                                                                    \
        static DevOnce once_;                                                               \
        if (!reserve_lds_once(once_, (const void*)(fn), (bytes))) {                         \
            set_error("%s: cannot reserve %d bytes of LDS", who, (int)(bytes));             \
            return STAR_ERR_LAUNCH;                                                         \
        }                                                                                   \
    } while (0)

template <int HD, int NG, bool DROP>
static int launch_fwd_ng(const void* q, const void* k, const void* v, const float* bias, const int32_t* key_len,
                         const star_xattn_shape* s, void* o, float* lse, hipStream_t st) {
    const int smem = FwdLds<HD, NG>::BYTES;
    XA_RESERVE((xattn_fwd_kernel<HD, NG, DROP>), smem, "star_xattn_fwd");
    dim3 grid(cdiv(s->L, 128), s->h, s->B);
    hipLaunchKernelGGL((xattn_fwd_kernel<HD, NG, DROP>), grid, dim3(NG * 256), smem, st, (const __bf16*)q,
                       (const __bf16*)k, (const __bf16*)v, bias, key_len, *s, (__bf16*)o, lse);
    return STAR_OK;
}

template <int HD, bool HAS_BIAS, bool DROP>
static int launch_fwd_ring(const void* q, const void* k, const void* v, const float* bias, const int32_t* key_len,
                           const star_xattn_shape* s, void* o, float* lse, hipStream_t st) {
    const int smem = Ring<HD>::BYTES;
    XA_RESERVE((xattn_fwd_ring_kernel<HD, HAS_BIAS, DROP>), smem, "star_xattn_fwd");
    // without a bias the DMA of the bias slot reads `lse` (any valid floats: never used)
    hipLaunchKernelGGL((xattn_fwd_ring_kernel<HD, HAS_BIAS, DROP>), dim3(cdiv(s->L, 128), s->h, s->B), dim3(FR_THREADS), smem,
                       st, (const __bf16*)q, (const __bf16*)k, (const __bf16*)v, bias, (const float*)lse, key_len, *s,
                       (__bf16*)o, lse);
    return STAR_OK;
}

template <int HD, bool DROP>
static int launch_fwd(const void* q, const void* k, const void* v, const float* bias, const int32_t* key_len,
                      const star_xattn_shape* s, void* o, float* lse, float* p_out, hipStream_t st) {
    int rc = STAR_OK;
    bool done = false;
    if constexpr (HD == 64 || HD == 128) {
        if (test_override(STAR_OVR_XA_FWD) != 1) {        // 1 = the register-staged kernel (test-only switch)
            rc = (bias && s->bias_len > 0) ? launch_fwd_ring<HD, true, DROP>(q, k, v, bias, key_len, s, o, lse, st)
                                           : launch_fwd_ring<HD, false, DROP>(q, k, v, bias, key_len, s, o, lse, st);
            done = true;
        }
    }
    if (!done) {
        // more than one 64-key tile: split the keys over two wave groups (two waves per SIMD)
        const int force = test_override(STAR_OVR_XA_NG);
        const bool two = force ? (force == 2) : (s->Kn > KT);
        rc = two ? launch_fwd_ng<HD, 2, DROP>(q, k, v, bias, key_len, s, o, lse, st)
                 : launch_fwd_ng<HD, 1, DROP>(q, k, v, bias, key_len, s, o, lse, st);
    }
    if (rc) return rc;
    if (p_out) {
        const int64_t rows = (int64_t)s->B * s->h * s->L;
        hipLaunchKernelGGL((xattn_probs_kernel<HD, DROP>), dim3(cdiv(rows, 4)), dim3(XA_THREADS), 0, st,
                           (const __bf16*)q, (const __bf16*)k, bias, key_len, (const float*)lse, *s, p_out);
    }
    return STAR_OK;
}

template <int HD, int NG, bool DROP>
static int launch_bwd_ng(const void* q, const void* k, const void* v, const void* o, const void* do_,
                         const float* lse, const float* bias, const int32_t* key_len, const star_xattn_shape* s,
                         void* dq, void* dk, void* dv, float* delta, float* dbias_part, hipStream_t st) {
    const int smem_dq = DqLds<HD, NG>::BYTES;
    // test-only switches (star_test_override): force a kernel variant for A/B runs and coverage
    const bool dq_gen = test_override(STAR_OVR_XA_DQ) == 1;
    const bool dkv_gen = test_override(STAR_OVR_XA_DKV) == 1;
    const int bwd_force = test_override(STAR_OVR_XA_BWD);      // 1 = never the fused kernel, 2 = whenever legal
    if constexpr (HD == 64 || HD == 128) {
        const bool fused_ok = s->L <= BwdFused<HD>::QROWS && !dq_gen && !dkv_gen;
        // measured at configs[1]'s shape from B = 1 to 128: the fused kernel wins at every batch size (K = 375: 3 rounds);
        // only few (batch, head) pairs with MANY key rounds are left to the split pair, whose dK/dV kernel spreads
        // the key groups of one pair over several workgroups
        const int64_t bh_f = (int64_t)s->B * s->h;
        const bool want = bwd_force ? (bwd_force >= 2) : (bh_f >= 32 || s->Kn <= 512);
        if constexpr (HD == 128) {
            // head dim 128, test override 3 only: the 4-wave form (32 keys per wave, 32x32x16 MFMAs, half the LDS bytes
            // per FLOP; xattn_bwd4.hip).  Measured SLOWER than the 8-wave kernel at configs[1] (55 against 43 us, cold
            // caches): with one wave per SIMD nothing overlaps its softmax arithmetic with its MFMAs
            if (fused_ok && bwd_force == 3)
                return launch_bwd_fused4(DROP, q, k, v, o, do_, lse, bias, key_len, s, dq, dk, dv, dbias_part, st);
        }
        if (fused_ok && want) {
            const int smem_f = BwdFused<HD>::BYTES;
            XA_RESERVE((xattn_bwd_fused_small_kernel<HD, DROP>), smem_f, "star_xattn_bwd");
            hipLaunchKernelGGL((xattn_bwd_fused_small_kernel<HD, DROP>), dim3(1, s->h, s->B), dim3(512), smem_f, st,
                               (const __bf16*)q, (const __bf16*)k, (const __bf16*)v, (const __bf16*)o,
                               (const __bf16*)do_, lse, bias, key_len, *s, (__bf16*)dq, (__bf16*)dk, (__bf16*)dv,
                               dbias_part);
            return STAR_OK;
        }
    }
    bool dq_done = false;
    if constexpr (HD == 64 || HD == 128) {
        if (!dq_gen) {
            const int smem_s = Ring<HD>::BYTES;
            const bool hb = bias && s->bias_len > 0;
            // without a bias the DMA of the bias slot reads `lse` (any valid floats: never used)
            if (hb) {
                XA_RESERVE((xattn_bwd_dq_ring_kernel<HD, true, DROP>), smem_s, "star_xattn_bwd");
                hipLaunchKernelGGL((xattn_bwd_dq_ring_kernel<HD, true, DROP>), dim3(cdiv(s->L, 128), s->h, s->B),
                                   dim3(512), smem_s, st, (const __bf16*)q, (const __bf16*)k, (const __bf16*)v,
                                   (const __bf16*)o, (const __bf16*)do_, lse, delta, bias, (const float*)lse, key_len,
                                   *s, (__bf16*)dq);
            } else {
                XA_RESERVE((xattn_bwd_dq_ring_kernel<HD, false, DROP>), smem_s, "star_xattn_bwd");
                hipLaunchKernelGGL((xattn_bwd_dq_ring_kernel<HD, false, DROP>), dim3(cdiv(s->L, 128), s->h, s->B),
                                   dim3(512), smem_s, st, (const __bf16*)q, (const __bf16*)k, (const __bf16*)v,
                                   (const __bf16*)o, (const __bf16*)do_, lse, delta, bias, (const float*)lse, key_len,
                                   *s, (__bf16*)dq);
            }
            dq_done = true;
        }
    }
    if (!dq_done) {
        XA_RESERVE((xattn_bwd_dq_kernel<HD, NG, DROP>), smem_dq, "star_xattn_bwd");
        hipLaunchKernelGGL((xattn_bwd_dq_kernel<HD, NG, DROP>), dim3(cdiv(s->L, 128), s->h, s->B), dim3(NG * 256),
                           smem_dq, st, (const __bf16*)q, (const __bf16*)k, (const __bf16*)v, (const __bf16*)o,
                           (const __bf16*)do_, lse, delta, bias, key_len, *s, (__bf16*)dq);
    }
    if constexpr (HD == 64 || HD == 128) {
        if (s->L <= 256 && !dkv_gen) {
            // one workgroup (8 independent waves) per (batch, head); split the key groups over more workgroups
            // only when B*h alone cannot fill the chip
            const int ngroups = cdiv(s->Kn, 16);
            int nch = 1;
            const int64_t bh = (int64_t)s->B * s->h;
            if (bh < 192) nch = (int)min((int64_t)cdiv(ngroups, DkvSmall<HD>::WAVES), (256 + bh - 1) / bh);
            if (s->L <= 128) {
                const int smem_s = DkvSmall<HD, 128>::BYTES;
                XA_RESERVE((xattn_bwd_dkv_small_kernel<HD, DROP, 128>), smem_s, "star_xattn_bwd");
                hipLaunchKernelGGL((xattn_bwd_dkv_small_kernel<HD, DROP, 128>), dim3(nch, s->h, s->B), dim3(512), smem_s,
                                   st, (const __bf16*)q, (const __bf16*)k, (const __bf16*)v, (const __bf16*)do_, lse,
                                   (const float*)delta, bias, key_len, *s, (__bf16*)dk, (__bf16*)dv, dbias_part);
            } else {
                const int smem_s = DkvSmall<HD, 256>::BYTES;
                XA_RESERVE((xattn_bwd_dkv_small_kernel<HD, DROP, 256>), smem_s, "star_xattn_bwd");
                hipLaunchKernelGGL((xattn_bwd_dkv_small_kernel<HD, DROP, 256>), dim3(nch, s->h, s->B), dim3(512), smem_s,
                                   st, (const __bf16*)q, (const __bf16*)k, (const __bf16*)v, (const __bf16*)do_, lse,
                                   (const float*)delta, bias, key_len, *s, (__bf16*)dk, (__bf16*)dv, dbias_part);
            }
            return STAR_OK;
        }
    }
    const int smem_r = DkvRegLds<HD>::BYTES;
    XA_RESERVE((xattn_bwd_dkv_reg_kernel<HD, DROP>), smem_r, "star_xattn_bwd");
    hipLaunchKernelGGL((xattn_bwd_dkv_reg_kernel<HD, DROP>), dim3(cdiv(s->Kn, KB), s->h, s->B), dim3(256), smem_r, st,
                       (const __bf16*)q, (const __bf16*)k, (const __bf16*)v, (const __bf16*)do_, lse,
                       (const float*)delta, bias, key_len, *s, (__bf16*)dk, (__bf16*)dv, dbias_part);
    return STAR_OK;
}

template <int HD, bool DROP>
static int launch_bwd(const void* q, const void* k, const void* v, const void* o, const void* do_, const float* lse,
                      const float* bias, const int32_t* key_len, const star_xattn_shape* s, void* dq, void* dk,
                      void* dv, float* delta, float* dbias_part, float* dbias, int accumulate, hipStream_t st) {
    const int force = test_override(STAR_OVR_XA_NG);
    const bool two = force ? (force == 2) : true;
    int rc = two ? launch_bwd_ng<HD, 2, DROP>(q, k, v, o, do_, lse, bias, key_len, s, dq, dk, dv, delta, dbias_part, st)
                 : launch_bwd_ng<HD, 1, DROP>(q, k, v, o, do_, lse, bias, key_len, s, dq, dk, dv, delta, dbias_part, st);
    if (rc) return rc;
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
    const bool drop = s->dropout_p > 0.f;
#define XA_FWD(HD_) (drop ? launch_fwd<HD_, true>(q, k, v, bias, key_len, s, o, lse, p_out, st) \
                          : launch_fwd<HD_, false>(q, k, v, bias, key_len, s, o, lse, p_out, st))
    switch (s->hd) {
        case 64: rc = XA_FWD(64); break;
        case 96: rc = XA_FWD(96); break;
        default: rc = XA_FWD(128); break;
    }
#undef XA_FWD
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
    const bool drop = s->dropout_p > 0.f;
#define XA_BWD(HD_)                                                                                                \
    (drop ? launch_bwd<HD_, true>(q, k, v, o, do_, lse, bias, key_len, s, dq, dk, dv, delta, dbias_part, dbias,     \
                                  accumulate, st)                                                                  \
          : launch_bwd<HD_, false>(q, k, v, o, do_, lse, bias, key_len, s, dq, dk, dv, delta, dbias_part, dbias,    \
                                   accumulate, st))
    switch (s->hd) {
        case 64: rc = XA_BWD(64); break;
        case 96: rc = XA_BWD(96); break;
        default: rc = XA_BWD(128); break;
    }
#undef XA_BWD
    if (rc) return rc;
    STAR_LAUNCH_CHECK("star_xattn_bwd");
    return STAR_OK;
}

extern "C" int star_xattn_dbias_sum(const float* const* parts_host, int nl, int B, int h, int Kn, int bias_len,
                                    float* dbias, void* stream) {
    using namespace star;
    STAR_REQUIRE(parts_host && dbias, "star_xattn_dbias_sum: null pointer");
    STAR_REQUIRE(nl >= 1 && nl <= DBIAS_MAX_LAYERS, "star_xattn_dbias_sum: %d layers not in [1, %d]", nl, DBIAS_MAX_LAYERS);
    STAR_REQUIRE(B >= 1 && B <= 65535 && h >= 1 && Kn >= 1 && bias_len >= 1 && bias_len <= Kn,
                 "star_xattn_dbias_sum: bad shape B=%d h=%d Kn=%d bias_len=%d", B, h, Kn, bias_len);
    DbiasParts parts = {};
    for (int l = 0; l < nl; ++l) {
        STAR_REQUIRE(parts_host[l], "star_xattn_dbias_sum: parts[%d] is null", l);
        parts.p[l] = parts_host[l];
    }
    hipLaunchKernelGGL(xattn_dbias_sum_kernel, dim3(cdiv(bias_len, 64), B), dim3(256), 0, (hipStream_t)stream, parts, nl,
                       B, h, Kn, bias_len, dbias);
    STAR_LAUNCH_CHECK("star_xattn_dbias_sum");
    return STAR_OK;
}

#ifdef STAR_XA_STAMP
// diagnostic builds only: copy the in-kernel time stamps (1024 workgroups x 16 slots) to the host
extern "C" int star_debug_read_stamps(unsigned long long* out_host) {
    return hipMemcpyFromSymbol(out_host, HIP_SYMBOL(star::xa_stamps), sizeof(unsigned long long) * 1024 * 16) == hipSuccess ? 0 : -2;
}
#endif
