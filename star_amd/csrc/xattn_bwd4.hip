// Score-biased cross-attention backward, short targets, head dim 128: the 4-wave form of the fused kernel
// (reference autograd of MyGPT2Attention._attn, src/models/my_transformer.py:535-603).
//
// Same algorithm and the same ONE pass over K / V as xattn_bwd_fused_small_kernel (xattn.hip): a workgroup owns one
// (batch, head), keeps Q and dO (<= 128 queries) in LDS, walks the keys in rounds of 128, and per round runs
//   phase B  wave w owns 32 keys: S, dP, dS against every query; dK^T, dV^T in accumulators; parks its K rows and its
//            bf16 dS block [32 keys][128 queries] in LDS
//   barrier
//   phase C  wave w owns 32 queries: dQ^T += K^T dS^T over the round's 128 keys (both operands by transposed reads)
//   barrier, dK / dV of the round leave through the wave's own LDS rows as whole 16-byte-per-lane rows.
// What differs is the shape of the work, chosen for the LDS: the 8-wave kernel gives a wave 16 keys and 16x16x32 MFMAs,
// so every wave re-reads all of Q and dO (row-wise and transposed) per round -- 1 MB of LDS reads per 128 keys, which is
// what bounded its phase B (5.3 of a round's 7.8 us at configs[1]).  Here a wave is alone on its SIMD with the whole
// 512-register file, takes 32 keys and 32x32x16 MFMAs: half the LDS bytes per FLOP, no two-waves-per-SIMD hand-over, and registers
// to spare for whole 32 x 32 S / dP / P / dS tiles.
//
// MFMA 32x32x16 maps (lane l: r = l & 31, h = l >> 5): A[row r][k = 8h + j], B[k = 8h + j][col r], j = 0..7;
// C[row = (reg & 3) + 8 (reg >> 2) + 4h][col r].  S = Q K^T and dP = dO V^T are formed with the key on the lane
// (A = query rows from LDS, B = the wave's K / V fragment registers), so the P and dS accumulators, converted to bf16
// eight registers at a time, ARE the B operands of dV^T += dO^T P and dK^T += Q^T dS (k = query, in the permuted order
// 16 s + 8 (j >> 2) + 4 h + (j & 3), which the transposed reads of dO / Q reproduce).
// LDS images: 256-byte rows, 16-byte chunk c of row r stored at chunk position c ^ key(r),
// key(r) = ((r & 3) << 2) | ((r >> 2) & 3): conflict-free for the ds_read_b128 row reads AND the ds_read_b64_tr_b16
// transposed reads of the 32x32x16 operands.
#include "common.h"
#include "xattn_drop.h"

namespace star {

// ---- in-kernel time stamps (diagnostic builds only: -DSTAR_XA_STAMP, tools/build_stamp.sh; the product library has none)
#ifdef STAR_XA_STAMP
__device__ unsigned long long xa4_stamps[1024 * 16];
#define XA4_STAMP(slot)                                                                                            \
    do {                                                                                                           \
        if (threadIdx.x == 0)                                                                                      \
            xa4_stamps[((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) % 1024 * 16 + (slot)] =    \
                __builtin_amdgcn_s_memrealtime();                                                                  \
    } while (0)
#else
#define XA4_STAMP(slot) do {} while (0)
#endif

namespace {

constexpr float LOG2E_ = 1.4426950408889634f;

struct Bwd4 {
    static constexpr int HD = 128, QROWS = 128, KROWS = 128, WAVES = 4, ROWB = 256;
    static constexpr int Q_OFF = 0;
    static constexpr int G_OFF = Q_OFF + QROWS * ROWB;          // dO
    static constexpr int K_OFF = G_OFF + QROWS * ROWB;          // K rows of the round
    static constexpr int DS_OFF = K_OFF + KROWS * ROWB;         // dS^T of the round: [128 keys][128 queries] bf16
    static constexpr int LSE_OFF = DS_OFF + KROWS * ROWB;
    static constexpr int BYTES = LSE_OFF + 2 * QROWS * 4;       // 129 KiB
};

__device__ __forceinline__ int img_key(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }
__device__ __forceinline__ int img_off(int row, int ch) { return 256 * row + 16 * (ch ^ img_key(row)); }

__device__ __forceinline__ bf16x8 tr8(const uint8_t* p0, const uint8_t* p1) {
    bf16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, p0));
    bf16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, p1));
    return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
}

__device__ __forceinline__ void lds_barrier() {
    // LDS traffic of this wave done, then the workgroup barrier; global prefetches and stores stay in flight (a
    // __syncthreads() would wait for them: its fence covers global memory as well)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <bool DROP>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void xattn_bwd_fused4_kernel(
    const __bf16* __restrict__ Q, const __bf16* __restrict__ K, const __bf16* __restrict__ V,
    const __bf16* __restrict__ O, const __bf16* __restrict__ dO, const float* __restrict__ lse,
    const float* __restrict__ bias, const int32_t* __restrict__ key_len, star_xattn_shape s, __bf16* __restrict__ dQ,
    __bf16* __restrict__ dK, __bf16* __restrict__ dV, float* __restrict__ dbias_part) {
    using F = Bwd4;
    constexpr int KS = 8;             // k-steps of a 32x32x16 MFMA over the head dim
    constexpr int MB = 4;             // 32-row head-dim blocks of dK^T / dV^T / dQ^T
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t* q_lds = smem + F::Q_OFF;
    uint8_t* g_lds = smem + F::G_OFF;
    uint8_t* k_lds = smem + F::K_OFF;
    uint8_t* ds_lds = smem + F::DS_OFF;
    float* lse_lds = (float*)(smem + F::LSE_OFF);
    float* del_lds = lse_lds + F::QROWS;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hh = lane >> 5;
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
    const float scale2 = s.scale * LOG2E_;
    const Drop<DROP> drop(s, b, h);
    const int ngroups = (s.Kn + 31) >> 5;
    const int nrounds = (s.Kn + F::KROWS - 1) / F::KROWS;

    XA4_STAMP(0);
    // first round's K / V fragments (key on the lane: lane (r, hh) takes 16 bytes of key r per k-step), requested before
    // the Q / dO staging so that everything is in flight together
    bf16x8 kf[KS], vf[KS];
    float bias_cur = 0.f;
    {
        const int key = min(wid * 32 + r, s.Kn - 1);
        const __bf16* krow = Kb + (int64_t)key * s.k_sr + hh * 8;
        const __bf16* vrow = Vb + (int64_t)key * s.v_sr + hh * 8;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            kf[ks] = *(const bf16x8*)(krow + ks * 16);
            vf[ks] = *(const bf16x8*)(vrow + ks * 16);
        }
        if (bias && key < s.bias_len) bias_cur = bias[(int64_t)b * s.bias_len + key];
    }
    // ---- stage Q and dO (rows >= L: zeros), delta = rowsum(dO * O), lse in the log2 domain
    {
        u32x4 qv[8], gv[8], ov[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int c = tid + i * 256;
            const int row = c >> 4, ch = c & 15;
            const int rr = min(row, s.L - 1);
            qv[i] = *(const u32x4*)(Qb + (int64_t)rr * s.q_sr + ch * 8);
            gv[i] = *(const u32x4*)(dOb + (int64_t)rr * s.o_sr + ch * 8);
            ov[i] = *(const u32x4*)(Ob + (int64_t)rr * s.o_sr + ch * 8);
        }
        float lreg = 0.f;
        if (tid < F::QROWS) lreg = lse[((int64_t)b * s.h + h) * s.L + min(tid, s.L - 1)] * LOG2E_;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int c = tid + i * 256;
            const int row = c >> 4, ch = c & 15;
            const bool live = row < s.L;
            const u32x4 z = {0u, 0u, 0u, 0u};
            const int off = img_off(row, ch);
            *(u32x4*)(q_lds + off) = live ? qv[i] : z;
            *(u32x4*)(g_lds + off) = live ? gv[i] : z;
            const bf16x8 g8 = __builtin_bit_cast(bf16x8, gv[i]);
            const bf16x8 o8 = __builtin_bit_cast(bf16x8, ov[i]);
            float d = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) d += (float)g8[j] * (float)o8[j];
#pragma unroll
            for (int m = 1; m < 16; m <<= 1) d += __shfl_xor(d, m, 64);
            if (ch == 0) del_lds[row] = live ? d : 0.f;
        }
        if (tid < F::QROWS) lse_lds[tid] = tid < s.L ? lreg : 0.f;
    }
    __syncthreads();
    XA4_STAMP(1);

    const int nslices = (s.L + 31) >> 5;
    // per-lane address pieces (see the header comment)
    const int keyr = img_key(r);
    const int g4 = lane >> 4, i16 = lane & 15, qq = i16 >> 2, pp = i16 & 3;
    // row reads of the A operand: chunk 2 ks + hh of row (base + r)
    int roff[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) roff[ks] = 256 * r + 16 * ((2 * ks + hh) ^ keyr);
    // transposed reads, phase B (A = dO^T / Q^T of slice q0, head-dim block mb, k-step s2, half u):
    //   row = q0 + 16 s2 + 8 u + 4 (g4 >> 1) + qq, chunk = 4 mb + 2 (g4 & 1) + (pp >> 1), byte 8 (pp & 1)
    const int trb_base = 256 * (4 * (g4 >> 1) + qq) + 8 * (pp & 1);
    int trb[MB][2];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int u = 0; u < 2; ++u)
            trb[mb][u] = trb_base + 2048 * u +
                         16 * ((((mb ^ qq) & 3) << 2) | (((2 * (g4 & 1) + (pp >> 1)) ^ (2 * u + (g4 >> 1))) & 3));
    // transposed reads, phase C (A = K^T block mb / B = dS^T of this wave's queries; k-step s3 over the round's keys,
    // half u):  row = 16 s3 + 8 (g4 >> 1) + 4 u + qq, chunk = 4 (mb | wid) + 2 (g4 & 1) + (pp >> 1)
    const int trc_base = 256 * (8 * (g4 >> 1) + qq) + 8 * (pp & 1);
    int trc[MB][2];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int u = 0; u < 2; ++u)
            trc[mb][u] = trc_base + 1024 * u +
                         16 * ((((mb ^ qq) & 3) << 2) | (((2 * (g4 & 1) + (pp >> 1)) ^ (2 * (g4 >> 1) + u)) & 3));
    // ... and of B = dS^T: the 16-byte chunk that holds this wave's queries obeys the same formula with mb -> wid
    int trd[2];
#pragma unroll
    for (int u = 0; u < 2; ++u)
        trd[u] = trc_base + 1024 * u +
                 16 * ((((wid ^ qq) & 3) << 2) | (((2 * (g4 & 1) + (pp >> 1)) ^ (2 * (g4 >> 1) + u)) & 3));
    const int qw0 = wid * 32;                                     // phase C: first query of this wave
    const bool wave_has_q = qw0 < s.L;

    f32x16 dq[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int e = 0; e < 16; ++e) dq[mb][e] = 0.f;

#pragma unroll 1
    for (int rnd = 0; rnd < nrounds; ++rnd) {
        const int grp = rnd * F::WAVES + wid;
        const int key = grp * 32 + r;
        const bool key_ok = key < kend;
        const float my_bias2 = bias_cur * LOG2E_;
        const int qmin = key_ok ? (s.ilk_diag >= 0 ? key - s.ilk_diag : -(1 << 30)) : 0x7fffffff;
        // park this group's K rows for phase C (rows past Kn hold a clamped, finite row: their dS is zero)
        uint8_t* krow_lds = k_lds + 256 * (32 * wid);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) *(bf16x8*)(krow_lds + roff[ks]) = kf[ks];
        uint8_t* ds_row = ds_lds + 256 * (32 * wid + r) + 8 * hh;

        f32x16 dk[MB], dv[MB];
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int e = 0; e < 16; ++e) { dk[mb][e] = 0.f; dv[mb][e] = 0.f; }
        float db = 0.f;

#pragma unroll 1
        for (int sl = 0; sl < nslices; ++sl) {
            const int q0 = sl * 32;
            const uint8_t* qrow = q_lds + 256 * q0;
            const uint8_t* grow = g_lds + 256 * q0;
            // S[q][key], dP[q][key]: A = 32 query rows (row reads), B = the wave's K / V fragments
            // One wave per SIMD: nothing else hides an LDS round trip, so every fragment of a phase is requested
            // before the phase's first MFMA (the compiler left alone reads each fragment right before its use and
            // waits for it: 51 us per launch against 41 for the 8-wave kernel).
            f32x16 sacc, pacc;
#pragma unroll
            for (int e = 0; e < 16; ++e) { sacc[e] = 0.f; pacc[e] = 0.f; }
            // Fragment registers are recycled as soon as the MFMAs that read them have been ISSUED (an MFMA takes its
            // operands in its first cycles; the next reads land while those MFMAs execute): 32 registers of fragments
            // live at a time, not 64 (which spilled).
            {
                bf16x8 fa[KS];
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) fa[ks] = *(const bf16x8*)(qrow + roff[ks]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks], kf[ks], sacc, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) fa[ks] = *(const bf16x8*)(grow + roff[ks]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) pacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks], vf[ks], pacc, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            // transposed dO / Q fragments of the slice's first 16 queries: in flight during the softmax arithmetic
            const uint8_t* gtr = g_lds + 256 * q0;
            const uint8_t* qtr = q_lds + 256 * q0;
            bf16x8 gt[MB], qt[MB];
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                gt[mb] = tr8(gtr + trb[mb][0], gtr + trb[mb][1]);
                qt[mb] = tr8(qtr + trb[mb][0], qtr + trb[mb][1]);
            }
            __builtin_amdgcn_sched_barrier(0);
            // softmax arithmetic: register e = 4 i + j holds query q0 + 8 i + 4 hh + j of this lane's key
            f32x16 pv, dsv;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f32x4 ll = *(const f32x4*)(lse_lds + q0 + 8 * i + 4 * hh);
                const f32x4 dd = *(const f32x4*)(del_lds + q0 + 8 * i + 4 * hh);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int e = 4 * i + j;
                    const int qv_ = q0 + 8 * i + 4 * hh + j;
                    float p = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[e], scale2, my_bias2 - ll[j]));
                    const bool vis = (qv_ >= qmin) & (qv_ < s.L);
                    p = vis ? p : 0.f;
                    float ds;
                    if constexpr (DROP) {
                        const float km = drop.km(qv_, key);
                        ds = p * (pacc[e] * km - dd[j]);
                        p *= km;
                    } else {
                        ds = p * (pacc[e] - dd[j]);
                    }
                    db += ds;
                    pv[e] = p;
                    dsv[e] = ds;
                }
            }
            // bf16 P / dS: registers 8 s2 .. 8 s2 + 7 are the B fragment of k-step s2
            bf16x8 pf[2], df[2];
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    pf[s2][j] = (__bf16)pv[8 * s2 + j];
                    df[s2][j] = (__bf16)dsv[8 * s2 + j];
                }
            // dS^T of (this key, queries q0 + 8 i + 4 hh .. + 3) for phase C: chunk 4 sl + i of the key's row
            {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    bf16x4 d4;
#pragma unroll
                    for (int j = 0; j < 4; ++j) d4[j] = (i < 2) ? df[0][4 * i + j] : df[1][4 * (i - 2) + j];
                    *(bf16x4*)(ds_row + 16 * ((4 * sl + i) ^ keyr)) = d4;
                }
            }
            // dV^T += dO^T P, dK^T += Q^T dS: A = the transposed fragments requested above
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                dv[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gt[mb], pf[0], dv[mb], 0, 0, 0);
                dk[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qt[mb], df[0], dk[mb], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                gt[mb] = tr8(gtr + 4096 + trb[mb][0], gtr + 4096 + trb[mb][1]);
                qt[mb] = tr8(qtr + 4096 + trb[mb][0], qtr + 4096 + trb[mb][1]);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                dv[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gt[mb], pf[1], dv[mb], 0, 0, 0);
                dk[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qt[mb], df[1], dk[mb], 0, 0, 0);
            }
        }

        // The next round's K / V fragments go straight into the (now dead) fragment registers: in flight across the
        // barrier, phase C and the second barrier, and waited for BEFORE this round's stores are issued (vmcnt retires
        // in order).  (A second register set filled at the top of the round spilled: the arch VGPRs also hold the
        // S / dP / P / dS tiles that the softmax arithmetic works on.)
        {
            const int ng = grp + F::WAVES;
            const int nkey = min((ng < ngroups ? ng : grp) * 32 + r, s.Kn - 1);
            const __bf16* krow = Kb + (int64_t)nkey * s.k_sr + hh * 8;
            const __bf16* vrow = Vb + (int64_t)nkey * s.v_sr + hh * 8;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                kf[ks] = *(const bf16x8*)(krow + ks * 16);
                vf[ks] = *(const bf16x8*)(vrow + ks * 16);
            }
            bias_cur = 0.f;
            if (bias && nkey < s.bias_len) bias_cur = bias[(int64_t)b * s.bias_len + nkey];
        }

        XA4_STAMP(2 + 4 * (rnd & 3));
        lds_barrier();            // K rows and dS of the whole round are in LDS
        XA4_STAMP(3 + 4 * (rnd & 3));

        // phase C: this wave's 32 queries against the round's 128 keys
        if (wave_has_q) {
#pragma unroll 1
            for (int s3 = 0; s3 < F::KROWS / 16; s3 += 2) {
                const int k0 = rnd * F::KROWS + 16 * s3;
                if (k0 >= kend) break;                                                   // wave-uniform
                if (s.ilk_diag >= 0 && k0 > qw0 + 31 + s.ilk_diag) break;                // invisible to all 32 queries
                // two 16-key steps per iteration, all ten fragments requested before the first MFMA (keys past kend /
                // beyond the ILK band inside the pair carry dS = 0)
                const uint8_t* kb_ = k_lds + 4096 * s3;
                const uint8_t* db_ = ds_lds + 4096 * s3;
                bf16x8 dst[2], kt[2][MB];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    dst[t] = tr8(db_ + 4096 * t + trd[0], db_ + 4096 * t + trd[1]);
#pragma unroll
                    for (int mb = 0; mb < MB; ++mb) kt[t][mb] = tr8(kb_ + 4096 * t + trc[mb][0], kb_ + 4096 * t + trc[mb][1]);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int mb = 0; mb < MB; ++mb)
                        dq[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kt[t][mb], dst[t], dq[mb], 0, 0, 0);
            }
        }

        db += __shfl_xor(db, 32, 64);
        if (dbias_part && hh == 0 && key < s.Kn) dbias_part[((int64_t)b * s.h + h) * s.Kn + key] = db;

        // (the prefetched fragments are waited for here, before the stores below enter the same in-order queue)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(kf[ks]), "+v"(vf[ks]));
        asm volatile("" : "+v"(bias_cur));

        XA4_STAMP(4 + 4 * (rnd & 3));
        lds_barrier();            // phase C reads done: the round's LDS rows are free again
        XA4_STAMP(5 + 4 * (rnd & 3));

        // dK / dV of the round leave through the wave's own 32 rows of the K and dS tiles (free until the next round's
        // writes, which follow in program order): accumulator layout -> row-major, 16 bytes per lane
        {
            uint8_t* tk = k_lds + 256 * (32 * wid);
            uint8_t* tv = ds_lds + 256 * (32 * wid);
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    bf16x4 a, c;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        a[j] = (__bf16)(dk[mb][4 * i + j] * s.scale);
                        c[j] = (__bf16)dv[mb][4 * i + j];
                    }
                    // head-dim columns 32 mb + 8 i + 4 hh .. + 3 of key row r: chunk 4 mb + i, byte 8 hh
                    const int off = 256 * r + 16 * ((4 * mb + i) ^ (r & 15)) + 8 * hh;
                    *(bf16x4*)(tk + off) = a;
                    *(bf16x4*)(tv + off) = c;
                }
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int row = (lane >> 4) + 4 * it, c = lane & 15;
                const int off = 256 * row + 16 * (c ^ (row & 15));
                const u32x4 x = *(const u32x4*)(tk + off);
                const u32x4 y = *(const u32x4*)(tv + off);
                const int krow = grp * 32 + row;
                if (krow < s.Kn) {
                    *(u32x4*)(dKb + (int64_t)krow * s.k_sr + c * 8) = x;
                    *(u32x4*)(dVb + (int64_t)krow * s.v_sr + c * 8) = y;
                }
            }
        }
    }

    // dQ^T accumulators -> row-major through the wave's own rows (every wave passed the last barrier; the last round's
    // dK / dV copies above were read back by this same wave)
    {
        uint8_t* tq = k_lds + 256 * (32 * wid);
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                bf16x4 a;
#pragma unroll
                for (int j = 0; j < 4; ++j) a[j] = (__bf16)(dq[mb][4 * i + j] * s.scale);
                *(bf16x4*)(tq + 256 * r + 16 * ((4 * mb + i) ^ (r & 15)) + 8 * hh) = a;
            }
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int row = (lane >> 4) + 4 * it, c = lane & 15;
            const u32x4 x = *(const u32x4*)(tq + 256 * row + 16 * (c ^ (row & 15)));
            const int q = qw0 + row;
            if (q < s.L) *(u32x4*)(dQ + (int64_t)b * s.q_sb + (int64_t)h * s.q_sh + (int64_t)q * s.q_sr + c * 8) = x;
        }
    }
#ifdef STAR_XA_STAMP
    XA4_STAMP(14);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    XA4_STAMP(15);
#endif
}

}  // namespace

int launch_bwd_fused4(bool drop, const void* q, const void* k, const void* v, const void* o, const void* do_,
                      const float* lse, const float* bias, const int32_t* key_len, const star_xattn_shape* s, void* dq,
                      void* dk, void* dv, float* dbias_part, hipStream_t st) {
    static DevOnce once[2];
    const void* fn = drop ? (const void*)xattn_bwd_fused4_kernel<true> : (const void*)xattn_bwd_fused4_kernel<false>;
    if (!reserve_lds_once(once[drop ? 1 : 0], fn, Bwd4::BYTES)) {
        set_error("star_xattn_bwd: cannot reserve %d bytes of LDS", Bwd4::BYTES);
        return STAR_ERR_LAUNCH;
    }
    if (drop)
        hipLaunchKernelGGL(xattn_bwd_fused4_kernel<true>, dim3(1, s->h, s->B), dim3(256), Bwd4::BYTES, st,
                           (const __bf16*)q, (const __bf16*)k, (const __bf16*)v, (const __bf16*)o, (const __bf16*)do_, lse,
                           bias, key_len, *s, (__bf16*)dq, (__bf16*)dk, (__bf16*)dv, dbias_part);
    else
        hipLaunchKernelGGL(xattn_bwd_fused4_kernel<false>, dim3(1, s->h, s->B), dim3(256), Bwd4::BYTES, st,
                           (const __bf16*)q, (const __bf16*)k, (const __bf16*)v, (const __bf16*)o, (const __bf16*)do_, lse,
                           bias, key_len, *s, (__bf16*)dq, (__bf16*)dk, (__bf16*)dv, dbias_part);
    return STAR_OK;
}

}  // namespace star

#ifdef STAR_XA_STAMP
extern "C" int star_debug_read_stamps4(unsigned long long* out_host) {
    return hipMemcpyFromSymbol(out_host, HIP_SYMBOL(star::xa4_stamps), sizeof(unsigned long long) * 1024 * 16) == hipSuccess ? 0 : -2;
}
#endif
