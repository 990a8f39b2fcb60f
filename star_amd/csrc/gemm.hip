// Scorer MLP on bf16 MFMA.
//
// Reference: nugget_model = Sequential(Linear(d,d), ReLU, Linear(d,1)) + squeeze(-1)
// (src/models/w2v_asr.py:113-117,553; w2v_simul.py:144-148,410-411).  The reference materialises
// H = relu(X W1^T + b1) ([B,T,d]) and runs a second GEMV; here the second layer is the epilogue of
// the first GEMM, so H never leaves the registers:
//     scores[m] = b2 + sum_n w2[n] * relu( sum_k X[m,k] W1[n,k] + b1[n] )
//
// One "NT" GEMM template (C[M,N] = A[M,K] . B[N,K]^T, both operands K-contiguous) serves forward
// and backward through its epilogue:
//   EPI_SCORE   relu/bias/w2 row-dot  -> per-column-tile partial scores (+ the ReLU sign bits, 1 bit per element of H,
//               which are all the backward needs of H: there is no recompute GEMM)               (forward)
//   EPI_SLAB    fp32 split-K slab store                                      (U = G'^T X_sel, G' = ds * relu'(H))
//   EPI_ROWS    bf16 store with row scatter                                  (dX = G' (diag(w2) W1))
// U comes from gemm_tn_kernel, which reads G' and the listed rows of X as they lie
// (both row-major by the reduction index; fragments by transposing LDS reads) and so needs no G'^T / X_sel^T copies;
// EPI_SLAB on those copies remains for row counts whose row ids do not fit the TN kernel's LDS (> 114688 rows at 16 splits).
// Padded batches: the forward gathers the valid frames through a row list (star_scorer_fwd_compact) or skips the row
// tiles without a valid frame (star_scorer_fwd_ragged).
// Tile 256x256x64, 512 threads = 8 waves (2x4), each wave 128x64 = 8x4 mfma_f32_16x16x32_bf16 (128 accumulator
// registers).  Operands are staged with global_load_lds (16 B/lane, LDS image lane-linear) into two LDS buffers
// of four 16-KiB half-tiles each (128 KiB).  A K tile is consumed in four phases, one accumulator quadrant
// (64x32 per wave, 16 MFMAs) per phase; every phase reads the register sub-tile it needs, issues the LDS-DMA of
// ONE half-tile seven half-tiles ahead, and is fenced by raw s_barriers.  The two waves that share a SIMD
// (wave w and w+4, i.e. the two wave rows) run one barrier apart, so one of them is always in its MFMA block
// while the other issues its ds_reads and DMA pieces (an LDS-DMA piece costs its wave 100+ issue cycles).
// vmcnt is counted and waited once per K tile (three half-tiles stay in flight across it).  The XOR swizzle that
// makes the ds_read_b128 fragment reads conflict-free is applied to the per-lane SOURCE address and to the read
// address (same involution).
// Workgroup -> tile mapping is XCD-aware: the column tiles of one row panel run on one XCD, so the
// A panel is fetched into one L2 instead of eight.
// Algorithmic work: forward 2*N_f*d*(d+1) FLOP; bytes: X 2*d*N_f + W1 2*d*d read, scores 4*N_f written.
#include "common.h"

#ifndef STAR_GEMM_ABL
#define STAR_GEMM_ABL 0      // diagnostic builds (tools/build_abl.sh): 1 no DMA in the K loop, 2 no ds_reads, 4 no epilogue,
                             // 8 every tile reads one of 8 cache-hot A panels, 32 (4-wave kernel) no staging stores; TN kernel: 1, 2, 64 no slab store (16, one
                             // ds_read_b128 in place of each pair of transposing reads, existed in the builtin version)
#endif

namespace star {

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int GEMM_THREADS = 512;                 // 8 waves: 2 (M) x 4 (N), each 128 x 64
constexpr int NWAVE = GEMM_THREADS / 64;
constexpr int TM = 8, TN = 4;                     // 16x16 accumulator tiles per wave
constexpr int ROW_BYTES = BK * 2;                 // 128-byte LDS rows (8 chunks of 16 B)
constexpr int HALF_BYTES = 128 * ROW_BYTES;       // half-tile: 128 operand rows x 64 k = 16 KiB
constexpr int BUF_BYTES = 4 * HALF_BYTES;         // B-half0 | A-half0 | B-half1 | A-half1
constexpr int GEMM_LDS = 2 * BUF_BYTES;           // 128 KiB -> one workgroup (2 waves/SIMD) per CU
constexpr int SCORE_BW_MAXN = 3072;               // b1 / w2 staged in LDS up to this many columns (24 KiB)
constexpr int SCORE_LDS = GEMM_LDS + 4096 + 2 * SCORE_BW_MAXN * 4;   // + row-reduction scratch and b1 / w2 behind the ring
// half-tile kinds in staging (= consumption) order.  A-half h holds, for both wave rows, the 64 rows of
// accumulator quadrant row h (tile rows wr*128 + h*64 ..); B-half h the 32 columns wc*64 + h*32 .. of all four
// wave columns.
enum { H_B0 = 0, H_A0 = 1, H_B1 = 2, H_A1 = 3 };
// epilogue staging (EPI_ROWS): every wave parks its 128 x 64 bf16 output tile TRANSPOSED in LDS
// ([64 n][128 m], pitch 272 B) so that global stores are 16 B per lane along whichever axis the consumer wants
constexpr int EP_PITCH = 128 * 2 + 16;
constexpr int EP_WAVE_BYTES = 64 * EP_PITCH;      // 17 KiB
constexpr int EP_RED_OFF = NWAVE * EP_WAVE_BYTES; // reduction scratch behind the eight wave tiles
constexpr int EPI_LDS = EP_RED_OFF + 8192;        // 144 KiB
static_assert(EPI_LDS >= GEMM_LDS && EPI_LDS <= 160 * 1024, "epilogue staging must cover the ring and fit the CU");

enum { EPI_SCORE = 0, EPI_SLAB = 2, EPI_ROWS = 3 };

// scorer backward reductions (scorer_bwd_finish_kernel), also runnable by extra workgroups of the dX GEMM launch
struct FinishParams {
    const float* slab;
    int splits, d;
    const __bf16* w1;
    const float* b1;
    const float* w2;
    const float* colpart;
    const float* dspart;
    int nblk;
    float* dw1;
    float* dw2;
    float* db1;
    float* db2;
};

constexpr int STAR_TN_MAX_BATCH = 16;
struct GemmParams {
    const __bf16* A;
    const __bf16* B;
    int64_t lda, ldb;
    int M, N;                 // logical sizes (rows of A / rows of B)
    int kchunks;              // K / BK in total
    int splits;               // split-K factor (EPI_SLAB)
    int no4;                  // EPI_SCORE: 1 = operands of 4 GiB or more (the 4-wave kernel addresses with 32-bit offsets)
    int gm, gn;               // XCD tile grouping (0, 0 = one row panel x all column tiles): see map_tile
    const int32_t* a_rows;    // optional row gather for A (M entries), rows >= a_rows_n clamp to entry 0
    int a_rows_n;
    // epilogue operands
    const float* b1;
    const float* w2;
    float* partial;           // EPI_SCORE: [nbn][Mpad]
    int64_t partial_ld;
    uint32_t* relu_bits;      // EPI_SCORE (optional): sign bits of H in accumulator order (relu_bit_addr below)
    const int32_t* live_tiles; // EPI_SCORE (optional): ascending ids of the row tiles to compute (ragged batches); the
    int n_live;               // other row tiles are skipped altogether
    float* slab;              // EPI_SLAB: [splits][M][N]
    __bf16* out;              // EPI_ROWS: out[out_rows[m]][n] (out_rows NULL: out[m][n], m < out_rows_n)
    int64_t ldo;
    const int32_t* out_rows;
    int out_rows_n;
    const float* out_bias;    // EPI_ROWS (optional): added per column before the bf16 rounding
    // EPI_ROWS (optional): the last fin_blocks workgroups of the launch do the scorer backward's reductions instead of
    // a GEMM tile (the dX GEMM at 12 k rows fills 188 of 256 CUs)
    int fin_blocks;
    FinishParams fin;
    // gemm_tn_kernel, batched form (star_gemm_tn_batched_bf16): tn_nbatch independent products of one shape, product i
    // on operands tn_a[i] / tn_b[i], each over tn_ksplit ranges of r; "split" s = i * tn_ksplit + range.  0 = off.
    int tn_nbatch, tn_ksplit;
    const __bf16* tn_a[STAR_TN_MAX_BATCH];
    const __bf16* tn_b[STAR_TN_MAX_BATCH];
};

__device__ __forceinline__ void glds16(const void* g, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// LDS image of a half-tile: [128 rows][8 chunks of 16 B]; chunk c of row r is stored at position c ^ swz8(r),
// swz8(r) = (r >> 1) & 7.  With this involution the 16 lanes that one ds_read_b128 cycle serves (rows 0-3,12-15
// of k-chunk q together with rows 4-11 of k-chunk q+1) fall on 16 distinct 16-B slots of the 256-B bank row.
// global_load_lds writes LDS lane-linearly, so the swizzle is applied to the per-lane SOURCE address when staging
// and to the address when reading (the same involution on both sides).
__device__ __forceinline__ int swz8(int row) { return (row >> 1) & 7; }

// rows of the wave's staged [n][m] tile -> GT[n][m] (16 B per lane, 256 B contiguous per row)
__device__ __forceinline__ void ep_store_nrows(const uint8_t* W, __bf16* __restrict__ dstT, int64_t ld, int n_base,
                                               int m_base, int n_limit, int lane) {
    const int chunk = lane & 15;
    const uint8_t* src = W + (lane >> 4) * EP_PITCH + chunk * 16;
    __bf16* dst = dstT + (int64_t)(n_base + (lane >> 4)) * ld + m_base + chunk * 8;     // running pointer
    const int rows_ok = n_limit - n_base - (lane >> 4);                                 // row it*4 valid iff < rows_ok
#pragma unroll
    for (int it = 0; it < 16; ++it) {
        const u32x4 v = *(const u32x4*)(src + it * 4 * EP_PITCH);
        if (it * 4 < rows_ok) *(u32x4*)dst = v;
        dst += 4 * ld;
        __builtin_amdgcn_sched_barrier(0);
    }
}

// transposed read of the staged tile: lane (i = lane & 15, g = lane >> 4) gets, for m = mb*16 + i, the eight
// consecutive n = nh*32 + 8g .. +7 (16 B), i.e. a piece of ROW m of the un-transposed tile
__device__ __forceinline__ u32x4 ep_read_mrow(const uint8_t* W, int mb, int nh, int lane) {
    const int i = lane & 15, g = lane >> 4;
    const uint8_t* p0 = W + (nh * 32 + 8 * g + (i >> 2)) * EP_PITCH + (mb * 16 + 4 * (i & 3)) * 2;
    bf16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, p0));
    bf16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, p0 + 4 * EP_PITCH));
    bf16x8 r = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
    return *(u32x4*)&r;
}

// EPI_SCORE epilogue of one 256 x 256 tile (shared by the two scorer-forward kernels): ReLU / bias / w2 row dot into
// per-column-tile partial scores, and the ReLU sign bits.  Persistent mode: the next tile's operands are landing in
// the ring right now, so the row reduction has its own LDS behind it and uses raw barriers (a __syncthreads() would
// drain the DMA queue).
// BW_STAGED: b1 / w2 come from the copy behind the ring.  They are read by inline asm: to hipcc a plain LDS load may
// alias the LDS-DMA in flight (the next tile's operands) and gets an s_waitcnt vmcnt(0) in front of it, and a runtime
// choice between the LDS copy and global loads ends in the same wait at the join -- either drains, at the start of
// every epilogue, the prologue that the persistent loop has just put in flight behind it.
template <bool BW_STAGED>
__device__ __forceinline__ void score_epilogue(f32x4 (&acc)[TM][TN], const GemmParams& p, uint8_t* lds,
                                               const float* bw_lds, int m_tile, int n_tile, int nbn) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 2, wc = wid & 3;
    const int ccol = lane & 15, crow4 = (lane >> 4) * 4;
    const int wrow0 = wr * 128, wcol0 = wc * 64;
    const int m0 = m_tile * BM, n0 = n_tile * BN;
    float* red = (float*)(lds + GEMM_LDS);
    float bb[TN], ww[TN];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        const int n = n0 + wcol0 + tn * 16 + ccol;
        if constexpr (BW_STAGED) {
            const uint32_t ad = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)(bw_lds + n);
            asm volatile("ds_read_b32 %0, %1" : "=v"(bb[tn]) : "v"(ad));
            asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(ww[tn]) : "v"(ad), "n"(SCORE_BW_MAXN * 4));
        } else {
            bb[tn] = n < p.N ? p.b1[n] : 0.f;
            ww[tn] = n < p.N ? p.w2[n] : 0.f;
        }
    }
    if constexpr (BW_STAGED) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    }
    // ReLU sign bits (all the backward needs of H) are packed per lane while the row dot is formed: element
    // e = (tm*4 + tn)*4 + j goes to bit 31 - e%32 of the lane's dword e/32 (v_alignbit shifts in the sign of -h)
    uint32_t hb[TM / 2] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
        float srow[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float h = acc[tm][tn][j] + bb[tn];
                srow[j] += fmaxf(h, 0.f) * ww[tn];
                hb[tm >> 1] = __builtin_amdgcn_alignbit(hb[tm >> 1], __float_as_uint(0.f - h), 31);
            }
        // sum over the 16 lanes that share a row: rotations by 8, 4, 2, 1 inside the DPP row leave the total in every
        // lane (four v_add_f32_dpp per value; __shfl_xor would be a chain of ds_bpermute round trips: 128 per tile,
        // each waited for, 6 us of the 32 us a tile took)
#define STAR_ROR_ADD(CTRL)                                                                                   \
    _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                            \
        srow[j] += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(srow[j]), (CTRL), 0xF, 0xF, false))
        STAR_ROR_ADD(0x128);
        STAR_ROR_ADD(0x124);
        STAR_ROR_ADD(0x122);
        STAR_ROR_ADD(0x121);
#undef STAR_ROR_ADD
        // lane ccol = j of each 16-lane group stores row j of the group's four rows
        const float v01 = ccol == 0 ? srow[0] : srow[1], v23 = ccol == 2 ? srow[2] : srow[3];
        if (ccol < 4) red[(wrow0 + tm * 16 + crow4 + ccol) * 4 + wc] = ccol < 2 ? v01 : v23;
    }
    if (p.relu_bits) {
        // tile blob: [wave 8][plane 4][lane 64] dwords -> every store instruction writes 256 contiguous bytes
        uint32_t* dst = p.relu_bits + ((int64_t)(m_tile * nbn + n_tile) * NWAVE + wid) * (TM / 2) * 64 + lane;
#pragma unroll
        for (int q = 0; q < TM / 2; ++q) dst[q * 64] = hb[q];
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);         // lgkmcnt(0): this wave's partial sums are in LDS
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    static_assert(BM <= GEMM_THREADS, "one thread per tile row");
    if (tid < BM) {
        const f32x4 rr = *(const f32x4*)(red + tid * 4);
        p.partial[(int64_t)n_tile * p.partial_ld + m0 + tid] = (rr[0] + rr[1]) + (rr[2] + rr[3]);
    }
    // (the next write to `red` is a whole K loop of barriers away)
}

// ---- scorer backward reductions for one output row n of W1, by a group of 256 threads (tid 0..255):
//   U[n][:] = sum of the split-K slabs;  c[n] = sum of the column partials;
//   dW1[n][:] = w2[n] * U[n][:]     db1[n] = w2[n] * c[n]     dw2[n] = <W1[n][:], U[n][:]> + b1[n] * c[n]
// (relu(h) = h * relu'(h), so sum_m ds[m] relu(h[m][n]) = <W1[n], U[n]> + b1[n] c[n]: no H needed);  db2 = sum ds.
// Fixed summation orders: bitwise reproducible.  Part a leaves the group's wave partials in red / red2 (4 floats each);
// the caller puts a workgroup barrier between part a and part b.
__device__ __forceinline__ void finish_row_a(const FinishParams& f, int n, int tid, float* red, float* red2) {
    const int lane = tid & 63, wid = tid >> 6;
    const int d = f.d;
    const int64_t dd = (int64_t)d * d;
    const float wn = f.w2[n];
    float dot = 0.f;
    for (int k = tid * 4; k < d; k += 1024) {
        const int64_t i = (int64_t)n * d + k;
        // four slabs per round: their loads are independent and in flight together; the order stays fixed
        f32x4 u = *(const f32x4*)(f.slab + i);
        int sidx = 1;
        for (; sidx + 3 < f.splits; sidx += 4) {
            const f32x4 v0 = *(const f32x4*)(f.slab + (int64_t)(sidx + 0) * dd + i);
            const f32x4 v1 = *(const f32x4*)(f.slab + (int64_t)(sidx + 1) * dd + i);
            const f32x4 v2 = *(const f32x4*)(f.slab + (int64_t)(sidx + 2) * dd + i);
            const f32x4 v3 = *(const f32x4*)(f.slab + (int64_t)(sidx + 3) * dd + i);
            u += v0;
            u += v1;
            u += v2;
            u += v3;
        }
        for (; sidx < f.splits; ++sidx) u += *(const f32x4*)(f.slab + (int64_t)sidx * dd + i);
        const bf16x4 wv = *(const bf16x4*)(f.w1 + i);
        dot += ((float)wv[0] * u[0] + (float)wv[1] * u[1]) + ((float)wv[2] * u[2] + (float)wv[3] * u[3]);
        *(f32x4*)(f.dw1 + i) = u * wn;
    }
    float c = 0.f;
    for (int t = tid; t < f.nblk; t += 256) c += f.colpart[(int64_t)t * d + n];
    dot = wave_sum(dot);
    c = wave_sum(c);
    if (lane == 0) {
        red[wid] = dot;
        red2[wid] = c;
    }
}
__device__ __forceinline__ void finish_row_b(const FinishParams& f, int n, int tid, const float* red, const float* red2) {
    const int lane = tid & 63, wid = tid >> 6;
    if (tid == 0) {
        const float dsum = (red[0] + red[1]) + (red[2] + red[3]);
        const float csum = (red2[0] + red2[1]) + (red2[2] + red2[3]);
        f.dw2[n] = dsum + f.b1[n] * csum;
        f.db1[n] = f.w2[n] * csum;
    }
    if (n == 0 && wid == 1) {                      // db2: one wave, lane-strided partial sums, then a fixed tree
        float s2 = 0.f;
        for (int t = lane; t < f.nblk; t += 64) s2 += f.dspart[t];
        s2 = wave_sum(s2);
        if (lane == 0) f.db2[0] = s2;
    }
}
// the same for a 512-thread workgroup `blk` of `nblocks`: its two halves take rows 2 i, 2 i + 1 of its share
__device__ __forceinline__ void finish_rows_block(const FinishParams& f, int blk, int nblocks, float* scratch) {
    const int tid = threadIdx.x, half = tid >> 8, t256 = tid & 255;
    float* red = scratch + half * 8;
    float* red2 = red + 4;
    const int per = ((f.d + nblocks - 1) / nblocks + 1) & ~1;              // rows per workgroup, even
    for (int i = 0; i < per; i += 2) {
        const int n = blk * per + i + half;
        const bool ok = n < f.d && i + half < per;
        if (ok) finish_row_a(f, n, t256, red, red2);
        __syncthreads();
        if (ok) finish_row_b(f, n, t256, red, red2);
        __syncthreads();
    }
}

// LIVE: the row tiles come from p.live_tiles (star_scorer_fwd_ragged); an instantiation of its own, so that the common
// kernels carry neither its loads nor its registers (with it the staged EPI_SCORE kernel spilled two VGPRs).
template <int EPI, bool BW_STAGED = true, bool LIVE = false>
__global__ __launch_bounds__(GEMM_THREADS) void gemm_nt_kernel(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    if constexpr (EPI == EPI_ROWS) {
        const int first_fin = (int)gridDim.x - p.fin_blocks;
        if ((int)blockIdx.x >= first_fin) {        // (fin_blocks = 0: never)
            finish_rows_block(p.fin, (int)blockIdx.x - first_fin, p.fin_blocks, (float*)lds);
            return;
        }
    }
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);     // wave-uniform
    const int wr = wid >> 2, wc = wid & 3;

    // ---- XCD-aware tile mapping: blocks b and b+8 share an XCD (round-robin dispatch); give every XCD whole
    // (row panel, K split) pairs so that all column tiles that share an A slab hit the same L2.
    // EPI_SCORE runs PERSISTENT: a workgroup walks tiles vb = blockIdx.x, + gridDim.x, ... (gridDim.x is a multiple
    // of 8, so it stays on its XCD) and the operand stream of the next tile is staged behind the last K tiles of the
    // current one, so only the first tile of a workgroup pays the pipeline prologue.
    constexpr bool PERSIST = (EPI == EPI_SCORE);
    const int nbm = (p.M + BM - 1) / BM, nbn = (p.N + BN - 1) / BN;
    // Grouping (non-persistent kernels with many column tiles): the tiles that an XCD runs at the same time (32, in
    // dispatch order) form gm row panels x gn column tiles, so that BOTH operand panels are shared through that XCD's
    // L2 -- with one row panel x all nbn column tiles (the default, right for nbn <= 8) an XCD streams the whole B
    // operand once per row panel.
    const int g_n = p.gn > 0 ? p.gn : nbn, g_m = p.gm > 0 ? p.gm : 1;
    auto map_tile = [&](int vb, int& mt, int& nt, int& sp) -> bool {
        const int xcd = vb & 7, local = vb >> 3;
        const int per_blk = g_m * g_n, nblk_n = (nbn + g_n - 1) / g_n;
        const int blk = local / per_blk, within = local - blk * per_blk;
        const int prow = blk / nblk_n, cblk = blk - prow * nblk_n;
        const int panel = (prow * g_m + within / g_n) * 8 + xcd;
        nt = cblk * g_n + within % g_n;
        if (nt >= nbn) return false;
        if constexpr (PERSIST && LIVE) {
            sp = 0;
            if (panel >= p.n_live) return false;
            mt = p.live_tiles[panel];
            return true;
        }
        mt = panel % nbm;
        sp = panel / nbm;
        return panel < nbm * p.splits;
    };
    int vb = blockIdx.x;
    int m_tile, n_tile, split;
    if (!map_tile(vb, m_tile, n_tile, split)) return;
    int m0 = m_tile * BM, n0 = n_tile * BN;

    const int cps = (p.kchunks + p.splits - 1) / p.splits;
    const int kc0 = split * cps;
    const int kc1 = min(p.kchunks, kc0 + cps);
    const int nk = max(0, kc1 - kc0);

    // ---- per-lane staging sources: wave w stages pieces 2w, 2w+1 (8 rows x 128 B each) of every half-tile;
    // lane l -> half-tile row piece*8 + (l >> 3), LDS position l & 7, source chunk (l & 7) ^ swz8(row).
    auto make_src = [&](const __bf16* (&out)[4][2], int mm0, int nn0) {
        // A rows first: with a row gather the four index loads are issued together (a load inside the pointer
        // loop would be waited for one at a time)
        int64_t arow[2][2];
#pragma unroll
        for (int hsel = 0; hsel < 2; ++hsel)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int hr = (wid * 2 + i) * 8 + (lane >> 3);
                int am = ((STAR_GEMM_ABL & 8) ? (mm0 & 2047) : mm0) + (hr >> 6) * 128 + hsel * 64 + (hr & 63);
                if (am >= p.M) am = p.M - 1;
                arow[hsel][i] = am;
            }
        if (p.a_rows) {
            int32_t g[2][2];
#pragma unroll
            for (int hsel = 0; hsel < 2; ++hsel)
#pragma unroll
                for (int i = 0; i < 2; ++i) g[hsel][i] = p.a_rows[arow[hsel][i] < p.a_rows_n ? arow[hsel][i] : 0];
#pragma unroll
            for (int hsel = 0; hsel < 2; ++hsel)
#pragma unroll
                for (int i = 0; i < 2; ++i) arow[hsel][i] = g[hsel][i];
        }
#pragma unroll
        for (int kind = 0; kind < 4; ++kind)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int hr = (wid * 2 + i) * 8 + (lane >> 3);
                const int chunk = (lane & 7) ^ swz8(hr);
                const int hsel = kind >> 1;
                if (kind & 1) {   // A operand
                    out[kind][i] = p.A + arow[hsel][i] * p.lda + (int64_t)kc0 * BK + chunk * 8;
                } else {          // B operand
                    int bn = nn0 + (hr >> 5) * 64 + hsel * 32 + (hr & 31);
                    if (bn >= p.N) bn = p.N - 1;
                    out[kind][i] = p.B + (int64_t)bn * p.ldb + (int64_t)kc0 * BK + chunk * 8;
                }
            }
    };
    const __bf16* src[4][2];
    const __bf16* nsrc[4][2];
    make_src(src, m0, n0);

    f32x4 acc[TM][TN];

    // stage half-tile KIND of K tile TT of the current output tile; K tiles past the end belong to the next
    // output tile of this workgroup (persistent mode).  UB = number of K tiles consumed so far: buffer parity.
#define STAR_STAGE(KIND, TT)                                                                                 \
    do {                                                                                                     \
        if ((STAR_GEMM_ABL & 1) && in_loop) break;                                                           \
        uint8_t* dst_ = lds + ((ubase + (TT)) & 1) * BUF_BYTES + (KIND) * HALF_BYTES + wid * 2048;           \
        if ((TT) < nk) {                                                                                     \
            glds16(src[KIND][0] + (int64_t)(TT) * BK, dst_);                                                 \
            glds16(src[KIND][1] + (int64_t)(TT) * BK, dst_ + 1024);                                          \
        } else if (has_next) {                                                                               \
            glds16(nsrc[KIND][0] + (int64_t)((TT) - nk) * BK, dst_);                                         \
            glds16(nsrc[KIND][1] + (int64_t)((TT) - nk) * BK, dst_ + 1024);                                  \
        }                                                                                                    \
    } while (0)

    // fragment reads: lane (frow = l & 15, fq = l >> 4) takes k-chunk 4*ks + fq of half-tile row (base + frow);
    // swz8(base + frow) == swz8(frow) because every base is a multiple of 16
    const int frow = lane & 15, fq = lane >> 4;
    const int choff0 = ((fq ^ swz8(frow)) << 4), choff1 = (((4 + fq) ^ swz8(frow)) << 4);
    const int a_off = (wr * 64 + frow) * ROW_BYTES;
    const int b_off = (wc * 32 + frow) * ROW_BYTES;
    bf16x8 af[4][2], bfr[2][2][2];      // A quadrant rows (reloaded per row half), B both column halves
    if (STAR_GEMM_ABL & 2) {
#pragma unroll
        for (int i = 0; i < 4; ++i) af[i][0] = af[i][1] = (bf16x8){1, 1, 1, 1, 1, 1, 1, 1};
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) bfr[i][j][0] = bfr[i][j][1] = (bf16x8){1, 1, 1, 1, 1, 1, 1, 1};
    }
    auto read_a = [&](int buf, int mh) {
        if (STAR_GEMM_ABL & 2) return;
        const uint8_t* base = lds + buf * BUF_BYTES + (mh ? H_A1 : H_A0) * HALF_BYTES + a_off;
        // k-step-major: the MFMA block starts with the ks = 0 products, whose fragments are then the first to arrive
#pragma unroll
        for (int i = 0; i < 4; ++i) af[i][0] = *(const bf16x8*)(base + i * 16 * ROW_BYTES + choff0);
#pragma unroll
        for (int i = 0; i < 4; ++i) af[i][1] = *(const bf16x8*)(base + i * 16 * ROW_BYTES + choff1);
    };
    auto read_b0 = [&](int buf) {
        if (STAR_GEMM_ABL & 2) return;
        const uint8_t* base = lds + buf * BUF_BYTES + H_B0 * HALF_BYTES + b_off;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            bfr[0][j][0] = *(const bf16x8*)(base + j * 16 * ROW_BYTES + choff0);
            bfr[0][j][1] = *(const bf16x8*)(base + j * 16 * ROW_BYTES + choff1);
        }
    };
    auto read_b1 = [&](int buf) {
        if (STAR_GEMM_ABL & 2) return;
        const uint8_t* base = lds + buf * BUF_BYTES + H_B1 * HALF_BYTES + b_off;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            bfr[1][j][0] = *(const bf16x8*)(base + j * 16 * ROW_BYTES + choff0);
            bfr[1][j][1] = *(const bf16x8*)(base + j * 16 * ROW_BYTES + choff1);
        }
    };
#define STAR_MMA(MH, NH)                                                                                     \
    do {                                                                                                     \
        __builtin_amdgcn_s_barrier();                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
        __builtin_amdgcn_s_setprio(1);                                                                       \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                     \
            _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                    \
                _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                \
                    acc[4 * (MH) + i][2 * (NH) + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(               \
                        af[i][ks], bfr[NH][j][ks], acc[4 * (MH) + i][2 * (NH) + j], 0, 0, 0);                \
        __builtin_amdgcn_s_setprio(0);                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
        __builtin_amdgcn_s_barrier();                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
    } while (0)

    // ---- prologue: all of tile 0 and three half-tiles of tile 1 (seven half-tiles ahead, as in the loop)
    int ubase = 0;
    bool has_next = false;
    bool in_loop = false;
    (void)in_loop;
    STAR_STAGE(H_B0, 0);
    STAR_STAGE(H_A0, 0);
    STAR_STAGE(H_B1, 0);
    STAR_STAGE(H_A1, 0);
    if (nk > 1) {
        STAR_STAGE(H_B0, 1);
        STAR_STAGE(H_A0, 1);
        STAR_STAGE(H_B1, 1);
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);

    // EPI_SCORE: b1 / w2 of all columns into LDS once, so that the epilogues (which run while the next tile's DMA
    // is in flight) touch no register-destination global load -- hipcc would drain vmcnt(0) at its first use
    float* bw_lds = (float*)(lds + GEMM_LDS + 4096);
    if constexpr (PERSIST) {
        if constexpr (BW_STAGED) {
            for (int n = tid; n < nbn * BN; n += GEMM_THREADS) {
                bw_lds[n] = n < p.N ? p.b1[n] : 0.f;
                bw_lds[SCORE_BW_MAXN + n] = n < p.N ? p.w2[n] : 0.f;
            }
            __syncthreads();
        }
    }

  for (;;) {      // output tiles of this workgroup (one iteration unless PERSIST)
    int nm_tile = 0, nn_tile = 0, nsplit = 0;
    if constexpr (PERSIST) {
        has_next = (nk >= 2) && map_tile(vb + (int)gridDim.x, nm_tile, nn_tile, nsplit);
        if (has_next) make_src(nsrc, nm_tile * BM, nn_tile * BN);
    }
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    if (wr == 1) __builtin_amdgcn_s_barrier();      // the second wave row runs one barrier behind the first
    __builtin_amdgcn_sched_barrier(0);

    in_loop = true;
    for (int t = 0; t < nk; ++t) {
        const int buf = (ubase + t) & 1;
        // phase 1: quadrant (0,0).  The four B reads go first so that lgkmcnt(8) retires them before the barrier:
        // their half-tile slot is re-staged in the next phase.
        read_b0(buf);
        __builtin_amdgcn_sched_barrier(0);
        read_a(buf, 0);
        STAR_STAGE(H_A1, t + 1);
        __builtin_amdgcn_s_waitcnt(0xC87F);         // lgkmcnt(8)
        STAR_MMA(0, 0);
        // phase 2: quadrant (0,1)
        read_b1(buf);
        STAR_STAGE(H_B0, t + 2);
        STAR_MMA(0, 1);
        // phase 3: quadrant (1,1)
        read_a(buf, 1);
        STAR_STAGE(H_A0, t + 2);
        STAR_MMA(1, 1);
        // phase 4: quadrant (1,0); retire everything up to the end of K tile t+1 (three half-tiles stay in flight)
        STAR_STAGE(H_B1, t + 2);
        if (t + 2 < nk || has_next) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        STAR_MMA(1, 0);
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();      // pairs with the second wave row's last barrier
    __builtin_amdgcn_sched_barrier(0);
#undef STAR_STAGE
#undef STAR_MMA
    if constexpr (!PERSIST) __syncthreads();  // LDS is reused by the epilogues below (no DMA in flight any more)

    // C layout of mfma_f32_16x16x32: col = lane & 15, row = (lane >> 4) * 4 + j
    const int ccol = lane & 15, crow4 = (lane >> 4) * 4;
    const int wrow0 = wr * 128, wcol0 = wc * 64;      // this wave's corner inside the tile

    if constexpr (EPI == EPI_SCORE && (STAR_GEMM_ABL & 4)) {
        float keep = 0.f;                                  // keep every accumulator alive
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) keep += (acc[tm][tn][0] + acc[tm][tn][1]) + (acc[tm][tn][2] + acc[tm][tn][3]);
        if (keep == 123.456f) p.partial[tid] = keep;
    } else if constexpr (EPI == EPI_SCORE) {
        score_epilogue<BW_STAGED>(acc, p, lds, bw_lds, m_tile, n_tile, nbn);
    } else if constexpr (EPI == EPI_SLAB) {
        float* slab = p.slab + (int64_t)split * p.M * p.N;
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int m = m0 + wrow0 + tm * 16 + crow4 + j;
                    const int n = n0 + wcol0 + tn * 16 + ccol;
                    if (m < p.M && n < p.N) slab[(int64_t)m * p.N + n] = acc[tm][tn][j];
                }
    } else {  // EPI_ROWS
        // destination rows of this wave's eight 16-row blocks: all eight index loads are issued here, together (a load
        // inside the store loop is waited for with vmcnt(0), which also waits for the stores of the block before it:
        // eight serialised round trips, 41 -> see DESIGN.md)
        int32_t orow_[8];
#pragma unroll
        for (int mb = 0; mb < 8; ++mb) {
            const int m = m0 + wrow0 + mb * 16 + (lane & 15);
            orow_[mb] = p.out_rows ? p.out_rows[min(m, p.out_rows_n - 1)] : m;
        }
        float ob[TN];
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
            const int n = n0 + wcol0 + tn * 16 + ccol;
            ob[tn] = (p.out_bias && n < p.N) ? p.out_bias[n] : 0.f;
        }
        uint8_t* W = lds + wid * EP_WAVE_BYTES;
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
                bf16x4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = (__bf16)(acc[tm][tn][j] + ob[tn]);
                *(bf16x4*)(W + (tn * 16 + ccol) * EP_PITCH + (tm * 16 + crow4) * 2) = o;
            }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (n0 + wcol0 < p.N) {
#pragma unroll
            for (int mb = 0; mb < 8; ++mb) {
                const int m = m0 + wrow0 + mb * 16 + (lane & 15);
                const bool ok = m < p.out_rows_n;
                const int64_t orow = orow_[mb];
#pragma unroll
                for (int nh = 0; nh < 2; ++nh) {
                    const u32x4 v = ep_read_mrow(W, mb, nh, lane);     // every lane takes part in the LDS read
                    if (ok) *(u32x4*)(p.out + orow * p.ldo + n0 + wcol0 + nh * 32 + 8 * (lane >> 4)) = v;
                }
            }
        }
    }

    if constexpr (!PERSIST) {
        break;
    } else {
        if (!has_next) break;
        vb += (int)gridDim.x;
        m_tile = nm_tile;
        n_tile = nn_tile;
        m0 = m_tile * BM;
        n0 = n_tile * BN;
#pragma unroll
        for (int kind = 0; kind < 4; ++kind) {
            src[kind][0] = nsrc[kind][0];
            src[kind][1] = nsrc[kind][1];
        }
        ubase += nk;
    }
  }
}

}  // namespace star
#include "gemm_nt4.h"
namespace star {

// ---------------------------------------------------------------------------------------------------------------
// "TN" GEMM for the scorer backward: U[n][k] = sum_r G'[r][n] * X[rows[r]][k] with BOTH operands stored row-major by
// the reduction index r (G' [Rpad][d] as the G' pass writes it, X gathered through rows[]): no G'^T, no X_sel^T pass.
// Same 256 x 256 x 64 tile, 8 waves, phases, barriers and LDS-DMA ring as gemm_nt_kernel; what differs:
//   * a half-tile is [64 r][128 columns] (256-byte rows); A-half h = columns m0 + 128 h .. + 127 of G', B-half h =
//     columns n0 + 128 h .. + 127 of X, so every staged row piece is 256 contiguous bytes.  Wave (wr, wc) owns output
//     rows {128 h + 64 wr + 0..63} and columns {128 h + 32 wc + 0..31}, h = 0, 1 (the four accumulator quadrants);
//   * MFMA fragments (8 consecutive r of one column per lane) come from ds_read_b64_tr_b16 (4 r x 16 columns per
//     16-lane group, two reads per fragment).  Conflict-free with the 32-byte windows of a row XOR-ed by
//     key(r) = (r & 3) | ((r >> 3) & 1) << 2: the 8 rows one half-wave reads fall into 8 different windows.  The DMA
//     applies the same involution to the source chunk;
//   * the row ids of the workgroup's r range are parked in LDS behind the ring before the loop (one ds_read per staged
//     row piece instead of a register-destination global load inside the counted-vmcnt pipeline).
constexpr int TN_MAX_ROWS = 7168;                                  // r range per workgroup (28 KiB of row ids)
constexpr int TN_LDS = GEMM_LDS + TN_MAX_ROWS * 4;
static_assert(TN_LDS <= 160 * 1024, "ring + row ids must fit the CU");
__device__ __forceinline__ int tn_key(int row) { return (row & 3) | (((row >> 3) & 1) << 2); }

__global__ __launch_bounds__(GEMM_THREADS) void gemm_tn_kernel(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 2, wc = wid & 3;
    const int nbm = (p.M + BM - 1) / BM, nbn = (p.N + BN - 1) / BN;
    // blocks b and b+8 share an XCD (round-robin dispatch): ALL output tiles of one r split run on one XCD, so the G'
    // and x rows of that split come out of HBM once and are shared through that XCD's L2 (every tile re-reads one of
    // the two operand slabs: with the tiles of a split spread over the XCDs x left HBM nbm times)
    const int vb = blockIdx.x;
    const int xcd = vb & 7, local = vb >> 3;
    const int tps = nbm * nbn;
    const int split = (local / tps) * 8 + xcd;
    if (split >= p.splits) return;
    const int m_tile = (local % tps) / nbn, n_tile = local % nbn;
    const int m0 = m_tile * BM, n0 = n_tile * BN;
    // batched form: split s = product s / tn_ksplit, r range s % tn_ksplit of that product's own operands
    const __bf16* Ap = p.A;
    const __bf16* Bp = p.B;
    int nsp = p.splits, sp = split;
    if (p.tn_nbatch > 0) {
        const int bi = split / p.tn_ksplit;
        sp = split - bi * p.tn_ksplit;
        nsp = p.tn_ksplit;
        // (constant indices: a run-time index into the by-value argument arrays would make hipcc copy them to scratch)
#pragma unroll
        for (int i = 0; i < STAR_TN_MAX_BATCH; ++i)
            if (bi == i) {
                Ap = p.tn_a[i];
                Bp = p.tn_b[i];
            }
    }
    const int cps = (p.kchunks + nsp - 1) / nsp;
    const int kc0 = sp * cps;
    const int kc1 = min(p.kchunks, kc0 + cps);
    const int nk = max(0, kc1 - kc0);

    // row ids of this workgroup's r range (rows past the list repeat entry 0: their G' rows are zero): parked in LDS
    // behind the ring -- after the prologue's DMA is on its way, see below.  No list (batched form): row r itself.
    int* rows_lds = (int*)(lds + GEMM_LDS);
    auto gathered = [&](int r) { return p.a_rows ? p.a_rows[r < p.a_rows_n ? r : 0] : min(r, p.a_rows_n - 1); };

    // staging geometry: wave w stages pieces 2w, 2w+1 (4 rows x 256 B each) of every half-tile; lane l -> row
    // piece*4 + (l >> 4), LDS chunk position l & 15, source chunk = the same position with its 32-byte window XOR-ed
    int srow[2], scol[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        srow[i] = (wid * 2 + i) * 4 + (lane >> 4);
        const int cpos = lane & 15;
        scol[i] = ((((cpos >> 1) ^ tn_key(srow[i])) << 1) | (cpos & 1)) * 8;          // element column 0..127
    }
    const __bf16* asrc[2][2];            // [half][piece]: G' + (kc0*64 + row) * lda + column
    int bcol[2][2];                      // [half][piece]: X column
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            int ca = m0 + h * 128 + scol[i];
            if (ca > p.M - 8) ca = p.M - 8;
            asrc[h][i] = Ap + ((int64_t)kc0 * BK + srow[i]) * p.lda + ca;
            int cb = n0 + h * 128 + scol[i];
            if (cb > p.N - 8) cb = p.N - 8;
            bcol[h][i] = cb;
        }

    f32x4 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

#define STAR_TN_STAGE_A(H, TT)                                                                               \
    do {                                                                                                     \
        if ((STAR_GEMM_ABL & 1) && tn_in_loop) break;                                                        \
        if ((TT) < nk) {                                                                                     \
            uint8_t* dst_ = lds + ((TT) & 1) * BUF_BYTES + ((H) ? H_A1 : H_A0) * HALF_BYTES + wid * 2048;     \
            glds16(asrc[H][0] + (int64_t)(TT) * BK * p.lda, dst_);                                           \
            glds16(asrc[H][1] + (int64_t)(TT) * BK * p.lda, dst_ + 1024);                                    \
        }                                                                                                    \
    } while (0)
#define STAR_TN_STAGE_B(H, TT, R0, R1)                                                                       \
    do {                                                                                                     \
        if ((STAR_GEMM_ABL & 1) && tn_in_loop) break;                                                        \
        if ((TT) < nk) {                                                                                     \
            uint8_t* dst_ = lds + ((TT) & 1) * BUF_BYTES + ((H) ? H_B1 : H_B0) * HALF_BYTES + wid * 2048;     \
            glds16(Bp + (int64_t)(R0) * p.ldb + bcol[H][0], dst_);                                           \
            glds16(Bp + (int64_t)(R1) * p.ldb + bcol[H][1], dst_ + 1024);                                    \
        }                                                                                                    \
    } while (0)
    auto row_id = [&](int tt, int i) { return rows_lds[min(tt, nk - 1) * BK + srow[i]]; };

    // fragment reads (see the header comment): lane (i16, g) reads row 8 g + (i16 >> 2) (+4), 8 bytes at 8 (i16 & 3)
    // of window (win ^ key); key is the same for both reads and both k steps
    const int i16 = lane & 15, g4 = lane >> 4;
    const int key = (i16 >> 2) | ((g4 & 1) << 2);
    const int rbase = (8 * g4 + (i16 >> 2)) * 256 + 8 * (i16 & 3);
    // The reads are inline asm: the ds_read_tr builtin has no memory operand, so next to an LDS-DMA in flight hipcc drains
    // vmcnt(0) before every group of them -- which emptied the seven-half-tile DMA pipeline four times per K tile (the
    // first version of this kernel: 123 us at R = 48000).  Asm reads are not tracked at all: the halves stay in
    // separate registers until STAR_TN_MMA has waited for lgkmcnt(0).
    bf16x4 afl[4][2], afh[4][2], bfl[2][2][2], bfh[2][2][2];
#define STAR_TN_TR(LO, HI, ADDR, OFF)                                                                        \
    do {                                                                                                     \
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(LO) : "v"(ADDR), "n"(OFF));                \
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(HI) : "v"(ADDR), "n"((OFF) + 4 * 256));    \
    } while (0)
    auto lds_addr32 = [](const uint8_t* q) {
        return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)q;
    };
    if (STAR_GEMM_ABL & 2) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) afl[i][ks] = afh[i][ks] = (bf16x4){1, 1, 1, 1};
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) bfl[i][j][ks] = bfh[i][j][ks] = (bf16x4){1, 1, 1, 1};
    }
    auto read_a = [&](int buf, int mh) {
        if (STAR_GEMM_ABL & 2) return;
        const uint8_t* base = lds + buf * BUF_BYTES + (mh ? H_A1 : H_A0) * HALF_BYTES + rbase;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t ad = lds_addr32(base + (((wr * 4 + i) ^ key) << 5));
            STAR_TN_TR(afl[i][0], afh[i][0], ad, 0);
            STAR_TN_TR(afl[i][1], afh[i][1], ad, 32 * 256);
        }
    };
    auto read_b = [&](int buf, int nh) {
        if (STAR_GEMM_ABL & 2) return;
        const uint8_t* base = lds + buf * BUF_BYTES + (nh ? H_B1 : H_B0) * HALF_BYTES + rbase;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const uint32_t ad = lds_addr32(base + (((wc * 2 + j) ^ key) << 5));
            STAR_TN_TR(bfl[nh][j][0], bfh[nh][j][0], ad, 0);
            STAR_TN_TR(bfl[nh][j][1], bfh[nh][j][1], ad, 32 * 256);
        }
    };
#define STAR_TN_F8(LO, HI) __builtin_shufflevector(LO, HI, 0, 1, 2, 3, 4, 5, 6, 7)
#define STAR_TN_MMA(MH, NH)                                                                                  \
    do {                                                                                                     \
        __builtin_amdgcn_s_barrier();                                                                        \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          /* the untracked fragment reads */       \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
        __builtin_amdgcn_s_setprio(1);                                                                       \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                     \
            _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                    \
                _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                \
                    acc[4 * (MH) + i][2 * (NH) + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(               \
                        STAR_TN_F8(afl[i][ks], afh[i][ks]), STAR_TN_F8(bfl[NH][j][ks], bfh[NH][j][ks]),      \
                        acc[4 * (MH) + i][2 * (NH) + j], 0, 0, 0);                                           \
        __builtin_amdgcn_s_setprio(0);                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
        __builtin_amdgcn_s_barrier();                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
    } while (0)

    bool tn_in_loop = false;
    (void)tn_in_loop;
    if (nk > 0) {
        // prologue: all of K tile 0 and three half-tiles of K tile 1 (seven half-tiles ahead, as in the loop).  The G'
        // halves of tile 0 go first (they need no row id); the row ids of tiles 0 and 1 are loaded by the staging
        // lanes themselves, so the first X pieces wait for one global load, not for an LDS round trip and a barrier.
        const int rr0 = kc0 * BK + srow[0], rr1 = kc0 * BK + srow[1];
        const int step1 = nk > 1 ? BK : 0;
        const int r0 = gathered(rr0), r1 = gathered(rr1), q0 = gathered(rr0 + step1), q1 = gathered(rr1 + step1);
        STAR_TN_STAGE_A(0, 0);
        STAR_TN_STAGE_A(1, 0);
        STAR_TN_STAGE_B(0, 0, r0, r1);
        STAR_TN_STAGE_B(1, 0, r0, r1);
        if (nk > 1) {
            STAR_TN_STAGE_B(0, 1, q0, q1);
            STAR_TN_STAGE_A(0, 1);
            STAR_TN_STAGE_B(1, 1, q0, q1);
        }
        for (int i = tid; i < nk * BK; i += GEMM_THREADS) rows_lds[i] = gathered(kc0 * BK + i);
        __syncthreads();            // row ids visible, K tile 0 (and the staged part of tile 1) landed

        if (wr == 1) __builtin_amdgcn_s_barrier();      // the second wave row runs one barrier behind the first
        __builtin_amdgcn_sched_barrier(0);
        tn_in_loop = true;
        for (int t = 0; t < nk; ++t) {
            const int buf = t & 1;
            const int n0r = row_id(t + 2, 0), n1r = row_id(t + 2, 1);     // rows of K tile t + 2
            // phase 1: quadrant (0,0).  The B reads go first: lgkmcnt(15) retires at least the first 9 of the 24 reads
            // in flight (all 8 of B-half 0) before the barrier; that half-tile slot is re-staged in the next phase.
            read_b(buf, 0);
            __builtin_amdgcn_sched_barrier(0);
            read_a(buf, 0);
            STAR_TN_STAGE_A(1, t + 1);
            __builtin_amdgcn_s_waitcnt(0xCF7F);         // lgkmcnt(15)
            STAR_TN_MMA(0, 0);
            // phase 2: quadrant (0,1)
            read_b(buf, 1);
            STAR_TN_STAGE_B(0, t + 2, n0r, n1r);
            STAR_TN_MMA(0, 1);
            // phase 3: quadrant (1,1)
            read_a(buf, 1);
            STAR_TN_STAGE_A(0, t + 2);
            STAR_TN_MMA(1, 1);
            // phase 4: quadrant (1,0)
            STAR_TN_STAGE_B(1, t + 2, n0r, n1r);
            if (t + 2 < nk) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            STAR_TN_MMA(1, 0);
        }
        if (wr == 0) __builtin_amdgcn_s_barrier();      // pairs with the second wave row's last barrier
        __builtin_amdgcn_sched_barrier(0);
    }
#undef STAR_TN_STAGE_A
#undef STAR_TN_STAGE_B
#undef STAR_TN_MMA
#undef STAR_TN_F8
#undef STAR_TN_TR

    // fp32 split slab: C layout of mfma_f32_16x16x32 is col = lane & 15, row = (lane >> 4) * 4 + j
    const int ccol = lane & 15, crow4 = (lane >> 4) * 4;
    float* slab = p.slab + (int64_t)split * p.M * p.N;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int m = m0 + (tm >> 2) * 128 + wr * 64 + (tm & 3) * 16 + crow4 + j;
                const int n = n0 + (tn >> 1) * 128 + wc * 32 + (tn & 1) * 16 + ccol;
                if ((STAR_GEMM_ABL & 64) ? acc[tm][tn][j] == 123.456f : (m < p.M && n < p.N))
                    slab[(int64_t)m * p.N + n] = acc[tm][tn][j];
            }
}

// scores[m] = b2 + sum over column tiles of partial[t][row(m)]   (fixed order -> reproducible)
// lengths (ragged forms): frame m = b*T + t is valid iff t < lengths[b]; the other frames get score 0 (their row may not
// have been computed at all).  offsets (compact form): the GEMM ran on the valid frames only, frame (b, t) is its row
// offsets[b] + t; without offsets row(m) = m.
__global__ void scorer_finish_kernel(const float* __restrict__ partial, int64_t ld, int ntiles,
                                     const float* __restrict__ b2, int64_t N, const int32_t* __restrict__ lengths,
                                     const int32_t* __restrict__ offsets, int T, float* __restrict__ scores) {
    int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= N) return;
    int64_t row = m;
    if (lengths) {
        const int b = (int)(m / T), t = (int)(m % T);
        if (t >= lengths[b]) {
            scores[m] = 0.f;
            return;
        }
        if (offsets) row = (int64_t)offsets[b] + t;
    }
    float s = b2[0];
    for (int t = 0; t < ntiles; ++t) s += partial[(int64_t)t * ld + row];
    scores[m] = s;
}

// 64x64 bf16 tile in LDS, pitch 144 B: rows are written 16 B at a time and read back column-wise (8 rows of one
// column -> one 16-B vector), so both the global reads and the global writes of a transpose are 16 B per lane
constexpr int TP_PITCH = 144;
__device__ __forceinline__ void tp_write_out(const uint8_t* tile, __bf16* __restrict__ dstT, int64_t ld, int i0, int r0) {
    // thread t: output rows ii = t/8 (+32), 8 consecutive r starting at (t%8)*8
    const int t = threadIdx.x, rb = (t & 7) * 8;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const int ii = (t >> 3) + pass * 32;
        bf16x8 v;
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = *(const __bf16*)(tile + (rb + k) * TP_PITCH + ii * 2);
        *(bf16x8*)(dstT + (int64_t)(i0 + ii) * ld + r0 + rb) = v;
    }
}

// XT[i][r] = X[rows[r]][i] for r < R, 0 for R <= r < Rpad   (bf16; d % 64 == 0, Rpad % 64 == 0)
__global__ __launch_bounds__(256) void gather_transpose_kernel(const __bf16* __restrict__ X, int64_t ldx,
                                                               const int32_t* __restrict__ rows, int R, int Rpad,
                                                               int d, __bf16* __restrict__ XT) {
    __shared__ __attribute__((aligned(16))) uint8_t tile[64 * TP_PITCH];
    const int r0 = blockIdx.x * 64, i0 = blockIdx.y * 64;
    const int t = threadIdx.x, c = t & 7;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const int rr = (t >> 3) + pass * 32;
        const int r = r0 + rr;
        u32x4 v = (u32x4){0u, 0u, 0u, 0u};
        if (r < R) v = *(const u32x4*)(X + (int64_t)rows[r] * ldx + i0 + c * 8);
        *(u32x4*)(tile + rr * TP_PITCH + c * 16) = v;
    }
    __syncthreads();
    tp_write_out(tile, XT, Rpad, i0, r0);
}

// dst[r][c] = bf16(src[r][c]);  dst_t[c][r] = bf16(src[r][c])
__global__ __launch_bounds__(256) void cast_transpose_generic_kernel(const float* __restrict__ src, int rows, int cols,
                                                             __bf16* __restrict__ dst, __bf16* __restrict__ dst_t,
                                                             const float* __restrict__ t_scale) {
    __shared__ float tile[64][65];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int rr = ty; rr < 64; rr += 4) {
        const int r = r0 + rr, c = c0 + tx;
        float v = (r < rows && c < cols) ? src[(int64_t)r * cols + c] : 0.f;
        if (dst && r < rows && c < cols) dst[(int64_t)r * cols + c] = (__bf16)v;
        if (t_scale && r < rows) v *= t_scale[r];
        tile[rr][tx] = v;
    }
    __syncthreads();
    if (dst_t)
        for (int cc = ty; cc < 64; cc += 4) {
            const int c = c0 + cc, r = r0 + tx;
            if (r < rows && c < cols) dst_t[(int64_t)c * rows + r] = (__bf16)tile[tx][cc];
        }
}

// same for rows % 64 == 0 and cols % 64 == 0: 16-byte accesses on every side
__global__ __launch_bounds__(256) void cast_transpose_kernel(const float* __restrict__ src, int rows, int cols,
                                                             __bf16* __restrict__ dst, __bf16* __restrict__ dst_t,
                                                             const float* __restrict__ t_scale) {
    __shared__ __attribute__((aligned(16))) uint8_t tile[64 * TP_PITCH];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int t = threadIdx.x, c = t & 7;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const int rr = (t >> 3) + pass * 32;
        const float* sp = src + (int64_t)(r0 + rr) * cols + c0 + c * 8;
        const f32x4 a = *(const f32x4*)sp, b = *(const f32x4*)(sp + 4);
        const float sc = t_scale ? t_scale[r0 + rr] : 1.f;
        bf16x8 v, vs;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            v[k] = (__bf16)a[k];
            v[4 + k] = (__bf16)b[k];
            vs[k] = (__bf16)(a[k] * sc);
            vs[4 + k] = (__bf16)(b[k] * sc);
        }
        if (dst) *(bf16x8*)(dst + (int64_t)(r0 + rr) * cols + c0 + c * 8) = v;
        *(bf16x8*)(tile + rr * TP_PITCH + c * 16) = vs;
    }
    __syncthreads();
    if (dst_t) tp_write_out(tile, dst_t, rows, c0, r0);
}

// Where the forward parks the sign bit of H[m][n] (accumulator order of the 256 x 256 tile, see the EPI_SCORE
// epilogue): dword index and bit position.  Row m: wr = (m>>7)&1, tm = (m>>4)&7, g = (m>>2)&3, j = m&3; column n:
// wc = (n>>6)&3, tn = (n>>4)&3, c = n&15.  The 16 dwords of (row, wave column) are contiguous (one per c).
__host__ __device__ static inline int64_t relu_bit_dword(int64_t m, int n, int nbn) {
    const int64_t tile = (m >> 8) * nbn + (n >> 8);
    const int wave = (int)((m >> 7) & 1) * 4 + ((n >> 6) & 3);
    const int plane = (int)((m >> 5) & 3), lane = (int)((m >> 2) & 3) * 16 + (n & 15);
    return ((tile * NWAVE + wave) * (TM / 2) + plane) * 64 + lane;
}
__host__ __device__ static inline int relu_bit_pos(int64_t m, int n) {
    return 31 - ((int)((m >> 4) & 1) * 16 + ((n >> 4) & 3) * 4 + (int)(m & 3));
}

// G' = ds * relu'(H) for the R listed rows, from the forward's sign bits: G' row-major (the A operand of the dX GEMM and
// of the TN dW1 GEMM; NULL skips it), G'^T (the A operand of the NT dW1 GEMM; NULL skips it), and per 64-row block the
// column sums c[n] = sum_r G'[r][n]
// (fp32 ds, not the bf16 copy) and sum_r ds[r].  One workgroup = 64 rows x GB_NCB blocks of 128 columns (a block = two
// wave columns of the forward's tile: 2 x 64 B of bits per row, staged in LDS with the row's dscore).  Writes
// Rpad * d * 2 bytes (twice that with G'^T), reads R * d / 8 * 8 (a row's bits share their 64-byte lines with 7
// other rows); bound by its two dependent round trips (row ids -> bits) per workgroup, not by the stores.
constexpr int GB_ROWS = 64, GB_COLS = 128, GB_PITCH = 33;      // LDS dwords per row (32 + 1: bank spread)
constexpr int GB_NCB = 2;                                      // column blocks per workgroup
__global__ __launch_bounds__(256) void scorer_bwd_g_kernel(const uint32_t* __restrict__ bits, int nbn,
                                                           const float* __restrict__ ds, const int32_t* __restrict__ rows,
                                                           const int32_t* __restrict__ bit_rows, int R, int Rpad, int d,
                                                           __bf16* __restrict__ G,
                                                           __bf16* __restrict__ GT, float* __restrict__ colpart,
                                                           float* __restrict__ dspart, int nblk,
                                                           u32x4* __restrict__ zfill, int64_t zfill_n16) {
    __shared__ uint32_t brow[GB_NCB * GB_ROWS * GB_PITCH];
    __shared__ float dsf[GB_ROWS];
    __shared__ int bsh[GB_ROWS];                   // bit position of (row, tn = 0); tn lowers it by 4
    const int tid = threadIdx.x;
    if ((int)blockIdx.x >= nblk) {
        // the workgroups past the row blocks zero-fill dx (store-only waves beside the issue-bound G' blocks: the fill
        // costs HBM write time this launch has to spare, and no second stream)
        const int64_t nz = (int64_t)(gridDim.x - nblk) * gridDim.y * 256;
        const u32x4 z = {0u, 0u, 0u, 0u};
        for (int64_t i = ((int64_t)(blockIdx.x - nblk) * gridDim.y + blockIdx.y) * 256 + tid; i < zfill_n16; i += nz)
            zfill[i] = z;
        return;
    }
    // GB_NCB column blocks of 128 per workgroup: one round trip for the row indices serves both, and twice the bytes
    // are in flight per resident workgroup (the pass is bound by two dependent round trips per workgroup at 8
    // workgroups per CU, not by its stores)
    const int r0 = blockIdx.x * GB_ROWS, n00 = blockIdx.y * (GB_NCB * GB_COLS);
    const int ncb = min(GB_NCB, (d - n00) / GB_COLS);
    // rows[r] = frame id (indexes ds); the forward's GEMM row of that frame (indexes the bits) is bit_rows[r] when the
    // forward ran on a compacted frame list, the frame id itself otherwise
    const int32_t* brows = bit_rows ? bit_rows : rows;
    if (tid < GB_ROWS) {
        const int r = r0 + tid, rc = min(r, R - 1);
        const float dv = ds[rows[rc]];
        dsf[tid] = r < R ? dv : 0.f;
        bsh[tid] = relu_bit_pos(brows[rc], 0);
    }
    {       // 128 B of bits per row and column block: thread -> (row, 16-byte piece p of 8): wave column p / 4, lanes
        // c = 4 (p % 4) .. +3.  All loads first, without a branch around them (rows past R read row R-1's bits: their ds
        // is 0; a column block past d re-reads the last one): one memory round trip for the index loads and one for the
        // bits
        const int row = tid >> 3, p8 = tid & 7;
        int64_t bm[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) bm[j] = brows[min(r0 + row + 32 * j, R - 1)];
        u32x4 v[GB_NCB][2];
#pragma unroll
        for (int cbk = 0; cbk < GB_NCB; ++cbk) {
            const int n0 = n00 + min(cbk, ncb - 1) * GB_COLS;
#pragma unroll
            for (int j = 0; j < 2; ++j)
                v[cbk][j] = *(const u32x4*)(bits + relu_bit_dword(bm[j], n0 + (p8 >> 2) * 64 + (p8 & 3) * 4, nbn));
        }
#pragma unroll
        for (int cbk = 0; cbk < GB_NCB; ++cbk)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                uint32_t* dstw = brow + (cbk * GB_ROWS + row + 32 * j) * GB_PITCH + p8 * 4;
                dstw[0] = v[cbk][j][0];
                dstw[1] = v[cbk][j][1];
                dstw[2] = v[cbk][j][2];
                dstw[3] = v[cbk][j][3];
            }
    }
    __syncthreads();
    // column nl (0..127) of a block: dword (nl >> 6) * 16 + (nl & 15) of the row, bit bsh[row] - 4 * ((nl >> 4) & 3)
    // pass 1: G' row-major, 16 B (8 columns) per thread and step.  A thread keeps its 8 columns over its 4 rows, so the
    // block's column sums (fp32 ds, not the bf16 copy) come from the same bit tests: per thread over its rows, then a
    // fixed xor tree over the 4 row groups of the wave, then the 4 waves in order through LDS.
    float csum[GB_NCB][8];
#pragma unroll
    for (int cbk = 0; cbk < GB_NCB; ++cbk) {
#pragma unroll
        for (int k = 0; k < 8; ++k) csum[cbk][k] = 0.f;
        if (cbk >= ncb) continue;
        const int n0 = n00 + cbk * GB_COLS;
        const uint32_t* bblk = brow + cbk * GB_ROWS * GB_PITCH;
#pragma unroll
        for (int i = 0; i < GB_ROWS * GB_COLS / 8 / 256; ++i) {
            const int c = tid + i * 256;
            const int row = c >> 4, cb = c & 15;                       // columns 8 cb .. 8 cb + 7
            const uint32_t* w = bblk + row * GB_PITCH + (cb >> 3) * 16 + (cb & 1) * 8;
            const int sh = bsh[row] - 4 * ((cb >> 1) & 3);
            const float dvf = dsf[row];
            const __bf16 dv = (__bf16)dvf;
            bf16x8 o;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const bool on = (w[k] >> sh) & 1u;
                o[k] = on ? dv : (__bf16)0.f;
                csum[cbk][k] += on ? dvf : 0.f;
            }
            if (G) *(bf16x8*)(G + (int64_t)(r0 + row) * d + n0 + cb * 8) = o;
        }
        // pass 2 (NT form only): G'^T, 16 B (8 rows of one column) per thread and step
        if (GT) {
#pragma unroll
            for (int i = 0; i < GB_ROWS * GB_COLS / 8 / 256; ++i) {
                const int c = tid + i * 256;
                const int nl = c >> 3, q8 = c & 7;
                const int wofs = (nl >> 6) * 16 + (nl & 15), tsh = 4 * ((nl >> 4) & 3);
                bf16x8 o;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int row = q8 * 8 + k;
                    const bool on = (bblk[row * GB_PITCH + wofs] >> (bsh[row] - tsh)) & 1u;
                    o[k] = on ? (__bf16)dsf[row] : (__bf16)0.f;
                }
                *(bf16x8*)(GT + (int64_t)(n0 + nl) * Rpad + r0 + q8 * 8) = o;
            }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            csum[cbk][k] += __shfl_xor(csum[cbk][k], 16, 64);
            csum[cbk][k] += __shfl_xor(csum[cbk][k], 32, 64);
        }
    }
    __syncthreads();                           // every wave is done with brow: its first words take the wave partials
    float* cpart = (float*)brow;               // [GB_NCB][4 waves][128 columns]
    if ((tid & 63) < 16) {
#pragma unroll
        for (int cbk = 0; cbk < GB_NCB; ++cbk)
#pragma unroll
            for (int k = 0; k < 8; ++k) cpart[(cbk * 4 + (tid >> 6)) * GB_COLS + (tid & 15) * 8 + k] = csum[cbk][k];
    }
    __syncthreads();
    for (int c = tid; c < ncb * GB_COLS; c += 256) {
        const float* cp = cpart + (c >> 7) * 4 * GB_COLS + (c & 127);
        colpart[(int64_t)blockIdx.x * d + n00 + c] = ((cp[0] + cp[GB_COLS]) + cp[2 * GB_COLS]) + cp[3 * GB_COLS];
    }
    if (blockIdx.y == 0 && tid < 64) {
        float v = dsf[tid];
        v = wave_sum(v);
        if (tid == 0) dspart[blockIdx.x] = v;
    }
}

// One workgroup per output row n of W1 (finish_row_a / _b above).
__global__ __launch_bounds__(256) void scorer_bwd_finish_kernel(FinishParams f) {
    __shared__ float red[4], red2[4];
    finish_row_a(f, blockIdx.x, threadIdx.x, red, red2);
    __syncthreads();
    finish_row_b(f, blockIdx.x, threadIdx.x, red, red2);
}

static int launch_gemm(int epi, const GemmParams& p, hipStream_t st) {
    const int nbm = (p.M + BM - 1) / BM, nbn = (p.N + BN - 1) / BN;
    // per XCD: its row panels in groups of gm, each group x ceil(nbn / gn) column blocks of gm * gn tiles (map_tile)
    const int g_n = p.gn > 0 ? p.gn : nbn, g_m = p.gm > 0 ? p.gm : 1;
    const int panels_xcd = ((p.live_tiles ? p.n_live : nbm * p.splits) + 7) / 8;
    int grid = 8 * ((panels_xcd + g_m - 1) / g_m) * ((nbn + g_n - 1) / g_n) * g_m * g_n;
    const bool bw_staged = nbn * BN <= SCORE_BW_MAXN;      // b1 / w2 fit behind the ring (d <= 3072)
    static DevOnce once[4], live_once;
    static std::atomic<int> cus[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) {
        set_error("scorer GEMM: hipGetDevice failed");
        return STAR_ERR_LAUNCH;
    }
    bool ok = true;
    const bool four = epi == EPI_SCORE && !p.no4 && p.kchunks >= 4 && p.kchunks % 2 == 0 && p.splits == 1 &&
                      test_override(STAR_OVR_SCORER_FWD) == 1;
    static DevOnce once4[6];
    void (*four_fn)(GemmParams) = nullptr;
    if (four) {
        const int v = (p.live_tiles ? 0 : bw_staged ? 1 : 2) * 2 + (p.relu_bits ? 1 : 0);
        switch (v) {
            case 0: four_fn = scorer_fwd4_kernel<false, true, false>; break;
            case 1: four_fn = scorer_fwd4_kernel<false, true, true>; break;
            case 2: four_fn = scorer_fwd4_kernel<true, false, false>; break;
            case 3: four_fn = scorer_fwd4_kernel<true, false, true>; break;
            case 4: four_fn = scorer_fwd4_kernel<false, false, false>; break;
            default: four_fn = scorer_fwd4_kernel<false, false, true>; break;
        }
        ok = reserve_lds_once(once4[v], (const void*)four_fn, G4_LDS);
    }
    switch (epi) {
        case EPI_SCORE:
            if (four) break;
            if (p.live_tiles)
                ok = reserve_lds_once(live_once, (const void*)gemm_nt_kernel<EPI_SCORE, false, true>, SCORE_LDS);
            else
                ok = bw_staged ? reserve_lds_once(once[0], (const void*)gemm_nt_kernel<EPI_SCORE, true>, SCORE_LDS)
                               : reserve_lds_once(once[1], (const void*)gemm_nt_kernel<EPI_SCORE, false>, SCORE_LDS);
            break;
        case EPI_SLAB: ok = reserve_lds_once(once[2], (const void*)gemm_nt_kernel<EPI_SLAB>, GEMM_LDS); break;
        default: ok = reserve_lds_once(once[3], (const void*)gemm_nt_kernel<EPI_ROWS>, EPI_LDS); break;
    }
    if (!ok) {
        set_error("scorer GEMM: cannot reserve the kernel's LDS (hipFuncSetAttribute failed)");
        return STAR_ERR_LAUNCH;
    }
    switch (epi) {
        case EPI_SCORE: {
            // persistent: one workgroup per CU (a multiple of 8 keeps every workgroup on its XCD) walks the tiles
            int n = cus[dev & 63].load(std::memory_order_relaxed);
            if (n == 0) {
                if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 8) n = 256;
                n = n / 8 * 8;
                cus[dev & 63].store(n, std::memory_order_relaxed);
            }
            if (grid > n) grid = n;
            if (four) {             // the experimental 4-wave kernel (gemm_nt4.h; measured slower): K
                                    // stages of 32 in groups of four, at least eight
                hipLaunchKernelGGL(four_fn, dim3(grid), dim3(G4_THREADS), G4_LDS, st, p);
                break;
            }
            if (p.live_tiles)      // (b1 / w2 by global loads: this form is kept for the ABI, not for speed)
                hipLaunchKernelGGL((gemm_nt_kernel<EPI_SCORE, false, true>), dim3(grid), dim3(GEMM_THREADS), SCORE_LDS, st, p);
            else if (bw_staged)
                hipLaunchKernelGGL((gemm_nt_kernel<EPI_SCORE, true>), dim3(grid), dim3(GEMM_THREADS), SCORE_LDS, st, p);
            else
                hipLaunchKernelGGL((gemm_nt_kernel<EPI_SCORE, false>), dim3(grid), dim3(GEMM_THREADS), SCORE_LDS, st, p);
            break;
        }
        case EPI_SLAB: hipLaunchKernelGGL(gemm_nt_kernel<EPI_SLAB>, dim3(grid), dim3(GEMM_THREADS), GEMM_LDS, st, p); break;
        default:
            hipLaunchKernelGGL(gemm_nt_kernel<EPI_ROWS>, dim3(grid + p.fin_blocks), dim3(GEMM_THREADS), EPI_LDS, st, p);
            break;
    }
    return STAR_OK;
}

static int launch_gemm_tn(const GemmParams& p, hipStream_t st) {
    const int nbm = (p.M + BM - 1) / BM, nbn = (p.N + BN - 1) / BN;
    const int grid = 8 * nbm * nbn * ((p.splits + 7) / 8);           // every XCD gets whole r splits
    if ((!p.a_rows && p.tn_nbatch < 1) || p.a_rows_n < 1) {
        set_error("scorer GEMM (TN): the row list of x is required");
        return STAR_ERR_ARG;
    }
    static DevOnce once;
    if (!reserve_lds_once(once, (const void*)gemm_tn_kernel, TN_LDS)) {
        set_error("scorer GEMM: cannot reserve the kernel's LDS (hipFuncSetAttribute failed)");
        return STAR_ERR_LAUNCH;
    }
    hipLaunchKernelGGL(gemm_tn_kernel, dim3(grid), dim3(GEMM_THREADS), TN_LDS, st, p);
    return STAR_OK;
}

static inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }
static inline int64_t round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }
static inline int bwd_splits(int d) {
    const int tiles = ((d + BM - 1) / BM) * ((d + BN - 1) / BN);
    int s = 512 / tiles;
    return s < 1 ? 1 : (s > 16 ? 16 : s);
}

}  // namespace star

// ---- scratch of star_scorer_fwd_static: the per-column-tile partial scores live in this array instead of a
// caller workspace.  One copy per GPU the library runs on, held until the process ends.  The name is a host-side
// stand-in, not a device address: host code only ever uses the per-device pointer from hipGetSymbolAddress.
namespace star {
constexpr size_t STATIC_SCRATCH_BYTES = STAR_SCORER_STATIC_SCRATCH_BYTES;
}
static __device__ __align__(256) unsigned char scratch[star::STATIC_SCRATCH_BYTES];
namespace star {
static std::atomic<void*> g_scratch_ptr[64];
// device address of `scratch` on the CURRENT device (looked up once per device, with that device current).  Every
// GPU holds its own copy of the array at its own address, so the pointer must belong to the device the kernels will
// run on: the stream's.  A stream of another device than the current one is refused (the library launches on the
// current device, as every other entry point does; a caller that switches devices makes the stream's device current).
static void* static_scratch(int* rc, hipStream_t stream) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) {
        set_error("star_scorer_fwd_static: cannot identify the current device");
        *rc = STAR_ERR_LAUNCH;
        return nullptr;
    }
    if (stream) {
        int sdev = dev;
        if (hipStreamGetDevice(stream, &sdev) != hipSuccess) {
            (void)hipGetLastError();
            set_error("star_scorer_fwd_static: cannot identify the stream's device");
            *rc = STAR_ERR_ARG;
            return nullptr;
        }
        if (sdev != dev) {
            set_error("star_scorer_fwd_static: the stream belongs to device %d but device %d is current: make the "
                      "stream's device current before the call", sdev, dev);
            *rc = STAR_ERR_ARG;
            return nullptr;
        }
    }
    void* ptr = g_scratch_ptr[dev].load(std::memory_order_acquire);
    if (!ptr) {
        hipError_t e = hipGetSymbolAddress((void**)&ptr, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !ptr) {
            set_error("star_scorer_fwd_static: hipGetSymbolAddress(scratch) failed: %s", hipGetErrorString(e));
            *rc = STAR_ERR_LAUNCH;
            return nullptr;
        }
        g_scratch_ptr[dev].store(ptr, std::memory_order_release);
    }
    *rc = STAR_OK;
    return ptr;
}
}  // namespace star

extern "C" size_t star_scorer_fwd_workspace(int64_t N, int d) {
    using namespace star;
    if (N <= 0 || d <= 0) return 0;
    return align256((size_t)((d + BN - 1) / BN) * (size_t)round_up(N, BM) * sizeof(float));
}

extern "C" size_t star_scorer_relu_bits_bytes(int64_t N, int d) {
    using namespace star;
    if (N <= 0 || d <= 0) return 0;
    // one 8-KiB blob per 256 x 256 output tile of the forward GEMM (column tiles padded to 256)
    return (size_t)(round_up(N, BM) / BM) * (size_t)((d + BN - 1) / BN) * (size_t)(BM * BN / 8);
}

namespace star {
static int scorer_fwd_impl(const char* who, const void* x, const void* w1, const float* b1, const float* w2,
                           const float* b2, int64_t N, int d, const int32_t* lengths, int T, const int32_t* live_tiles,
                           int n_live, const int32_t* offsets, const int32_t* rows, int64_t n_rows, float* scores,
                           void* relu_bits, void* workspace, size_t workspace_bytes, void* stream) {
    STAR_REQUIRE(x && w1 && b1 && w2 && b2 && scores && workspace, "%s: null pointer", who);
    STAR_REQUIRE(N >= 1 && N < (1ll << 31) - BM, "%s: bad N=%lld", who, (long long)N);
    STAR_REQUIRE(d >= 128 && d % 128 == 0, "%s: d=%d must be a multiple of 128", who, d);
    // M = the GEMM's rows: all N frames, or (compact form) the n_rows listed frames gathered through `rows`
    const int64_t M = rows ? n_rows : N;
    if (workspace_bytes < star_scorer_fwd_workspace(M, d)) {
        set_error("%s: workspace %zu < %zu bytes", who, workspace_bytes, star_scorer_fwd_workspace(M, d));
        return STAR_ERR_WORKSPACE;
    }
    GemmParams p = {};
    p.A = (const __bf16*)x;
    p.B = (const __bf16*)w1;
    p.lda = d;
    p.ldb = d;
    p.M = (int)M;
    p.a_rows = rows;
    p.a_rows_n = (int)(rows ? n_rows : 0);
    p.N = d;
    p.kchunks = d / BK;
    p.splits = 1;
    p.b1 = b1;
    p.w2 = w2;
    p.partial = (float*)workspace;
    p.partial_ld = round_up(M > 0 ? M : 1, BM);
    STAR_REQUIRE(((uintptr_t)relu_bits & 15) == 0, "%s: relu_bits must be 16-byte aligned", who);
    p.relu_bits = (uint32_t*)relu_bits;
    p.live_tiles = live_tiles;
    p.n_live = n_live;
    p.no4 = (N * (int64_t)d * 2 >= (1ll << 32)) || ((int64_t)d * d * 2 >= (1ll << 32));
    hipStream_t st = (hipStream_t)stream;
    if (M > 0 && (!live_tiles || n_live > 0))
        if (int rc = launch_gemm(EPI_SCORE, p, st)) return rc;
    hipLaunchKernelGGL(scorer_finish_kernel, dim3(cdiv(N, 256)), dim3(256), 0, st, (const float*)workspace,
                       p.partial_ld, (d + BN - 1) / BN, b2, N, lengths, offsets, T, scores);
    STAR_LAUNCH_CHECK(who);
    return STAR_OK;
}
}  // namespace star

extern "C" int star_scorer_fwd(const void* x, const void* w1, const float* b1, const float* w2, const float* b2,
                               int64_t N, int d, float* scores, void* relu_bits, void* workspace,
                               size_t workspace_bytes, void* stream) {
    return star::scorer_fwd_impl("star_scorer_fwd", x, w1, b1, w2, b2, N, d, nullptr, 1, nullptr, 0, nullptr, nullptr, 0,
                                 scores, relu_bits, workspace, workspace_bytes, stream);
}

extern "C" int star_scorer_fwd_ragged(const void* x, const void* w1, const float* b1, const float* w2, const float* b2,
                                      const int32_t* lengths, int B, int T, int d, const int32_t* live_tiles, int n_live,
                                      float* scores, void* relu_bits, void* workspace, size_t workspace_bytes,
                                      void* stream) {
    using namespace star;
    STAR_REQUIRE(lengths && live_tiles, "star_scorer_fwd_ragged: null pointer");
    STAR_REQUIRE(B >= 1 && T >= 1, "star_scorer_fwd_ragged: bad shape B=%d T=%d", B, T);
    const int64_t N = (int64_t)B * T;
    STAR_REQUIRE(n_live >= 0 && n_live <= (N + BM - 1) / BM, "star_scorer_fwd_ragged: n_live=%d for %lld row tiles", n_live,
                 (long long)((N + BM - 1) / BM));
    return scorer_fwd_impl("star_scorer_fwd_ragged", x, w1, b1, w2, b2, N, d, lengths, T, live_tiles, n_live, nullptr,
                           nullptr, 0, scores, relu_bits, workspace, workspace_bytes, stream);
}

extern "C" int star_scorer_fwd_compact(const void* x, const void* w1, const float* b1, const float* w2, const float* b2,
                                       const int32_t* lengths, const int32_t* offsets, const int32_t* rows,
                                       int64_t n_valid, int B, int T, int d, float* scores, void* relu_bits,
                                       void* workspace, size_t workspace_bytes, void* stream) {
    using namespace star;
    STAR_REQUIRE(lengths && offsets && rows, "star_scorer_fwd_compact: null pointer");
    STAR_REQUIRE(B >= 1 && T >= 1, "star_scorer_fwd_compact: bad shape B=%d T=%d", B, T);
    const int64_t N = (int64_t)B * T;
    STAR_REQUIRE(n_valid >= 0 && n_valid <= N, "star_scorer_fwd_compact: n_valid=%lld of %lld frames", (long long)n_valid,
                 (long long)N);
    return scorer_fwd_impl("star_scorer_fwd_compact", x, w1, b1, w2, b2, N, d, lengths, T, nullptr, 0, offsets, rows,
                           n_valid, scores, relu_bits, workspace, workspace_bytes, stream);
}

extern "C" int star_scorer_fwd_static(const void* x, const void* w1, const float* b1, const float* w2,
                                      const float* b2, int64_t N, int d, float* scores, void* relu_bits,
                                      void* stream) {
    using namespace star;
    STAR_REQUIRE(N >= 1 && d >= 128 && d % 128 == 0, "star_scorer_fwd_static: bad shape N=%lld d=%d", (long long)N, d);
    const size_t need = star_scorer_fwd_workspace(N, d);
    if (need > STATIC_SCRATCH_BYTES) {
        set_error("star_scorer_fwd_static: N=%lld x d=%d needs %zu bytes of scratch, the static array holds %zu "
                  "(use star_scorer_fwd with a caller workspace)", (long long)N, d, need, (size_t)STATIC_SCRATCH_BYTES);
        return STAR_ERR_WORKSPACE;
    }
    int rc = STAR_OK;
    void* ptr = static_scratch(&rc, (hipStream_t)stream);
    if (!ptr) return rc;
    // every partial the finish kernel reads is written by the GEMM of THIS call: nothing has to start at zero
    return star_scorer_fwd(x, w1, b1, w2, b2, N, d, scores, relu_bits, ptr, STATIC_SCRATCH_BYTES, stream);
}

namespace star {
struct BwdLayout {
    int64_t Rpad;
    int nblk, splits;
    size_t g_off, gt_off, xt_off, col_off, ds_off, slab_off, total;
};
// the TN dW1 GEMM parks the row ids of a workgroup's r range in LDS: usable while that range fits
// (measured, tools/tn_rows_check.py, 3 k - 48 k rows, scattered and contiguous: TN is 4-9 us faster up to 12 k rows,
// 20-60 us faster at 24 k - 48 k)
static bool bwd_use_tn(int64_t R, int d) {
    if (test_override(STAR_OVR_SCORER_NT) == 1) return false;      // 1 = NT (2 = TN: the default wherever it fits)
    const int64_t kchunks = round_up(R, BM) / BK;
    const int splits = bwd_splits(d);
    return (kchunks + splits - 1) / splits * BK <= TN_MAX_ROWS;
}
static BwdLayout bwd_layout(int64_t R, int d, bool want_dx) {
    BwdLayout L;
    L.Rpad = round_up(R, BM);
    L.nblk = (int)(L.Rpad / GB_ROWS);
    L.splits = bwd_splits(d);
    const size_t gbytes = align256((size_t)L.Rpad * d * 2);
    size_t off = 0;
    L.g_off = off;
    off += want_dx ? gbytes : 0;
    L.gt_off = off;
    off += gbytes;
    L.xt_off = off;
    off += gbytes;
    L.col_off = off;
    off += align256((size_t)L.nblk * d * 4);
    L.ds_off = off;
    off += align256((size_t)L.nblk * 4);
    L.slab_off = off;
    off += align256((size_t)L.splits * d * d * 4);
    L.total = off;
    return L;
}
}  // namespace star

extern "C" size_t star_scorer_bwd_workspace(int64_t R, int d) {
    using namespace star;
    if (R <= 0 || d <= 0) return 0;
    return bwd_layout(R, d, true).total;
}

extern "C" int star_scorer_bwd(const void* x, const void* w1, const void* w1ts, const float* b1, const float* w2,
                               const void* relu_bits, const float* dscores, const int32_t* rows, int64_t R, int64_t N,
                               int d, float* dw1, float* db1, float* dw2, float* db2, void* dx, int dx_zeroed,
                               void* workspace, size_t workspace_bytes, void* stream, void* aux_stream,
                               void* const* aux_events_host) {
    return star_scorer_bwd_mapped(x, w1, w1ts, b1, w2, relu_bits, dscores, rows, nullptr, R, N, d, dw1, db1, dw2, db2, dx,
                                  dx_zeroed, workspace, workspace_bytes, stream, aux_stream, aux_events_host);
}

extern "C" int star_scorer_bwd_mapped(const void* x, const void* w1, const void* w1ts, const float* b1, const float* w2,
                                      const void* relu_bits, const float* dscores, const int32_t* rows,
                                      const int32_t* bit_rows, int64_t R, int64_t N, int d, float* dw1, float* db1,
                                      float* dw2, float* db2, void* dx, int dx_zeroed, void* workspace,
                                      size_t workspace_bytes, void* stream, void* aux_stream,
                                      void* const* aux_events_host) {
    using namespace star;
    STAR_REQUIRE(x && w1 && b1 && w2 && relu_bits && dscores && rows && dw1 && db1 && dw2 && db2 && workspace,
                 "star_scorer_bwd: null pointer");
    STAR_REQUIRE(!dx || w1ts, "star_scorer_bwd: dx requested without w1ts");
    STAR_REQUIRE(R >= 1 && R <= N && N < (1ll << 31) - BM, "star_scorer_bwd: bad R=%lld N=%lld", (long long)R, (long long)N);
    STAR_REQUIRE(d >= 128 && d % 128 == 0, "star_scorer_bwd: d=%d must be a multiple of 128", d);
    STAR_REQUIRE(((uintptr_t)relu_bits & 15) == 0, "star_scorer_bwd: relu_bits must be 16-byte aligned");
    if (workspace_bytes < star_scorer_bwd_workspace(R, d)) {
        set_error("star_scorer_bwd: workspace %zu < %zu bytes", workspace_bytes, star_scorer_bwd_workspace(R, d));
        return STAR_ERR_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    // dW1 GEMM form: TN on the row-major G' and the gathered rows of x (no transposes), or -- for r ranges whose row
    // ids do not fit the kernel's LDS -- NT on G'^T and X_sel^T
    const bool tn = bwd_use_tn(R, d);
    const BwdLayout L = bwd_layout(R, d, dx != nullptr || tn);
    const int64_t Rpad = L.Rpad;
    uint8_t* ws = (uint8_t*)workspace;
    __bf16* G = (dx || tn) ? (__bf16*)(ws + L.g_off) : nullptr;
    __bf16* GT = tn ? nullptr : (__bf16*)(ws + L.gt_off);
    __bf16* XT = (__bf16*)(ws + L.xt_off);
    float* colpart = (float*)(ws + L.col_off);
    float* dspart = (float*)(ws + L.ds_off);
    float* slab = (float*)(ws + L.slab_off);

    // Two chains when the caller provides `aux_stream`: G' pass, dW1 GEMM and the slab / column reductions on `stream`;
    // X_sel^T (NT form), the zero-fill of dx and the dX GEMM on `aux_stream`.
    hipStream_t aux = aux_stream ? (hipStream_t)aux_stream : st;
    // fork / join events belong to the caller (one set per stream pair): the library keeps no event of its own
    hipEvent_t ev_fork[2] = {nullptr, nullptr}, ev_join[2] = {nullptr, nullptr};
    if (aux != st) {
        STAR_REQUIRE(aux_events_host && aux_events_host[0] && aux_events_host[1] && aux_events_host[2] &&
                         aux_events_host[3],
                     "star_scorer_bwd: aux_stream given without its four events");
        ev_fork[0] = (hipEvent_t)aux_events_host[0];
        ev_join[0] = (hipEvent_t)aux_events_host[1];
        ev_fork[1] = (hipEvent_t)aux_events_host[2];
        ev_join[1] = (hipEvent_t)aux_events_host[3];
    }
#define STAR_HIP(call)                                                                              \
    do {                                                                                            \
        hipError_t e_ = (call);                                                                     \
        if (e_ != hipSuccess) {                                                                     \
            set_error("star_scorer_bwd: %s failed: %s", #call, hipGetErrorString(e_));              \
            return STAR_ERR_LAUNCH;                                                                 \
        }                                                                                           \
    } while (0)
    // TN form: everything on `stream` (the aux stream and its events go unused)
    if (tn) aux = st;
    if (aux != st) {
        STAR_HIP(hipEventRecord(ev_fork[0], st));
        STAR_HIP(hipStreamWaitEvent(aux, ev_fork[0], 0));
    }
    // aux: X_sel^T for the NT form first (the dW1 GEMM waits for it), then the zero-fill of dx (rows that are not
    // listed), which only the dX GEMM -- issued on aux behind it -- needs
    if (!tn) {
        hipLaunchKernelGGL(gather_transpose_kernel, dim3((unsigned)(Rpad / 64), d / 64), dim3(256), 0, aux,
                           (const __bf16*)x, (int64_t)d, rows, (int)R, (int)Rpad, d, XT);
        if (aux != st) STAR_HIP(hipEventRecord(ev_join[0], aux));
        if (dx && !dx_zeroed) STAR_HIP(hipMemsetAsync(dx, 0, (size_t)N * d * 2, aux));
    }
    // 1. G' = ds * relu'(H) from the forward's sign bits: G' (for dX), G'^T (for dW1), column partials.  TN form: the
    // launch also carries the zero-fill of dx (about 1024 extra workgroups)
    static_assert(GB_COLS == 128, "d is a multiple of 128");
    const bool zfill = tn && dx && !dx_zeroed;
    if (zfill && ((uintptr_t)dx & 15)) {
        set_error("star_scorer_bwd: dx must be 16-byte aligned");
        return STAR_ERR_ARG;
    }
    const int gy = cdiv(d / GB_COLS, GB_NCB);
    const int zx = zfill ? (int)cdiv((int64_t)1024, (int64_t)gy) : 0;
    hipLaunchKernelGGL(scorer_bwd_g_kernel, dim3(L.nblk + zx, gy), dim3(256), 0, st,
                       (const uint32_t*)relu_bits, (d + BN - 1) / BN, dscores, rows, bit_rows, (int)R, (int)Rpad, d, G, GT,
                       colpart, dspart, L.nblk, zfill ? (u32x4*)dx : nullptr, zfill ? N * d / 8 : (int64_t)0);
    if (aux != st && !tn) STAR_HIP(hipStreamWaitEvent(st, ev_join[0], 0));     // X_sel^T is ready
    // 2. U = G'^T . X_sel  (split-K slabs)
    GemmParams q = {};
    q.M = d;
    q.N = d;
    q.kchunks = (int)(Rpad / BK);
    q.splits = L.splits;
    q.slab = slab;
    if (tn) {
        q.A = G;
        q.B = (const __bf16*)x;
        q.lda = d;
        q.ldb = d;
        q.a_rows = rows;            // here: the gather list of B's rows (r -> frame id)
        q.a_rows_n = (int)R;
        if (int rc = launch_gemm_tn(q, st)) return rc;
    } else {
        q.A = GT;
        q.B = XT;
        q.lda = Rpad;
        q.ldb = Rpad;
        if (int rc = launch_gemm(EPI_SLAB, q, st)) return rc;
    }
    // 3. + 4. dX rows = G' . (diag(w2) W1) and the reductions dW1 = diag(w2) U, dw2, db1, db2.
    FinishParams fin = {};
    fin.slab = slab;
    fin.splits = L.splits;
    fin.d = d;
    fin.w1 = (const __bf16*)w1;
    fin.b1 = b1;
    fin.w2 = w2;
    fin.colpart = colpart;
    fin.dspart = dspart;
    fin.nblk = L.nblk;
    fin.dw1 = dw1;
    fin.dw2 = dw2;
    fin.db1 = db1;
    fin.db2 = db2;
    GemmParams r = {};
    if (dx) {
        r.A = G;
        r.B = (const __bf16*)w1ts;
        r.lda = d;
        r.ldb = d;
        r.M = (int)Rpad;
        r.N = d;
        r.kchunks = d / BK;
        r.splits = 1;
        r.out = (__bf16*)dx;
        r.ldo = d;
        r.out_rows = rows;
        r.out_rows_n = (int)R;
    }
    if (dx && tn) {
        // TN form: ONE launch on the main stream -- the GEMM tiles plus 64 workgroups that do the reductions (at 12 k rows
        // the GEMM fills 188 CUs; a second stream for the reductions cost a 10-us cross-queue hand-over in the graph).
        // The zero-fill of dx rode in the G' launch.
        r.fin_blocks = 64;
        r.fin = fin;
        if (int rc = launch_gemm(EPI_ROWS, r, st)) return rc;
    } else {
        // NT form: dX GEMM on aux (behind the zero-fill of dx), the reductions on the main stream
        if (aux != st) {
            STAR_HIP(hipEventRecord(ev_fork[1], st));
            STAR_HIP(hipStreamWaitEvent(aux, ev_fork[1], 0));
        }
        if (dx)
            if (int rc = launch_gemm(EPI_ROWS, r, aux)) return rc;
        if (aux != st) STAR_HIP(hipEventRecord(ev_join[1], aux));
        hipLaunchKernelGGL(scorer_bwd_finish_kernel, dim3(d), dim3(256), 0, st, fin);
        if (aux != st) STAR_HIP(hipStreamWaitEvent(st, ev_join[1], 0));
    }
#undef STAR_HIP
    STAR_LAUNCH_CHECK("star_scorer_bwd");
    return STAR_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Row a9 / SURVEY 8f item 1: the cross-attention projections (HF Conv1D: y = x W + b, my_transformer.py:500-501,526).
// star_gemm_nt_bf16 is the 256 x 256 x 64 tile above as a plain projection GEMM: out = A . B^T + bias, bf16 in / out,
// fp32 accumulate, the bias added before the one rounding.  With N = nl * 2d = 8192 columns (the c_attn of all decoder
// layers in one launch) the tiles an XCD runs together are grouped 4 row panels x 8 column tiles, so both operand
// panels are shared through its L2.
extern "C" int star_gemm_nt_bf16(const void* a, int64_t lda, const void* b, int64_t ldb, const float* bias, int64_t M,
                                 int N, int K, void* out, int64_t ldo, void* stream) {
    using namespace star;
    STAR_REQUIRE(a && b && out, "star_gemm_nt_bf16: null pointer");
    STAR_REQUIRE(M >= 1 && M < (1ll << 31) - BM && N >= 64 && N % 64 == 0 && K >= 64 && K % 64 == 0,
                 "star_gemm_nt_bf16: bad shape M=%lld N=%d K=%d (N, K multiples of 64)", (long long)M, N, K);
    STAR_REQUIRE(lda >= K && ldb >= K && ldo >= N && lda % 8 == 0 && ldb % 8 == 0 && ldo % 8 == 0,
                 "star_gemm_nt_bf16: leading dimensions must cover the rows and be multiples of 8 elements");
    STAR_REQUIRE((((uintptr_t)a | (uintptr_t)b | (uintptr_t)out) & 15) == 0, "star_gemm_nt_bf16: operands must be 16-byte aligned");
    GemmParams r = {};
    r.A = (const __bf16*)a;
    r.B = (const __bf16*)b;
    r.lda = lda;
    r.ldb = ldb;
    r.M = (int)M;
    r.N = N;
    r.kchunks = K / BK;
    r.splits = 1;
    r.out = (__bf16*)out;
    r.ldo = ldo;
    r.out_rows = nullptr;
    r.out_rows_n = (int)M;
    r.out_bias = bias;
    const int nbn = (N + BN - 1) / BN;
    if (nbn > 8) {
        r.gn = 8;
        r.gm = 4;
    }
    if (int rc = launch_gemm(EPI_ROWS, r, (hipStream_t)stream)) return rc;
    STAR_LAUNCH_CHECK("star_gemm_nt_bf16");
    return STAR_OK;
}

// Weight gradients of several projections of one shape in ONE launch: out[i * ksplit + s][m][n] = sum over the s-th
// range of r of A_i[r][m] * B_i[r][n] (fp32; i < nbatch <= 16; both operands row-major by r, as the activations x and
// the output gradients g of an HF Conv1D lie: dW = x^T g with A = x, B = g).  One such product at M = N = 1024,
// R = 4096 has 16 tiles of 256 x 256 -- a sixteenth of the chip, and hipBLASLt answers it with 64 x 64 tiles at
// 0.28 PFLOP/s; eight of them, as the q_attn / c_proj layers of a 4-layer decoder produce per
// step, fill it.  a_list / b_list are HOST arrays of nbatch device pointers (they travel in the kernel arguments).
// The caller adds the ksplit partial products of a product up (fixed order).
extern "C" int star_gemm_tn_batched_bf16(const void* const* a_list, const void* const* b_list, int nbatch, int ksplit,
                                         int64_t R, int M, int N, int64_t lda, int64_t ldb, float* out, void* stream) {
    using namespace star;
    STAR_REQUIRE(a_list && b_list && out, "star_gemm_tn_batched_bf16: null pointer");
    STAR_REQUIRE(nbatch >= 1 && ksplit >= 1 && nbatch * ksplit <= STAR_TN_MAX_BATCH,
                 "star_gemm_tn_batched_bf16: nbatch * ksplit = %d * %d exceeds %d", nbatch, ksplit, STAR_TN_MAX_BATCH);
    STAR_REQUIRE(R >= BK && R % BK == 0 && (R / BK + ksplit - 1) / ksplit * BK <= TN_MAX_ROWS,
                 "star_gemm_tn_batched_bf16: R = %lld must be a multiple of %d with at most %d rows per range",
                 (long long)R, BK, TN_MAX_ROWS);
    STAR_REQUIRE(M >= 128 && N >= 128 && M % 8 == 0 && N % 8 == 0 && lda >= M && ldb >= N && lda % 8 == 0 && ldb % 8 == 0,
                 "star_gemm_tn_batched_bf16: bad shape M=%d N=%d lda=%lld ldb=%lld (>= 128, multiples of 8)", M, N,
                 (long long)lda, (long long)ldb);
    GemmParams r = {};
    for (int i = 0; i < nbatch; ++i) {
        STAR_REQUIRE(a_list[i] && b_list[i] && (((uintptr_t)a_list[i] | (uintptr_t)b_list[i]) & 15) == 0,
                     "star_gemm_tn_batched_bf16: operand %d is null or not 16-byte aligned", i);
        r.tn_a[i] = (const __bf16*)a_list[i];
        r.tn_b[i] = (const __bf16*)b_list[i];
    }
    r.A = r.tn_a[0];
    r.B = r.tn_b[0];
    r.lda = lda;
    r.ldb = ldb;
    r.M = M;
    r.N = N;
    r.kchunks = (int)(R / BK);
    r.splits = nbatch * ksplit;
    r.tn_nbatch = nbatch;
    r.tn_ksplit = ksplit;
    r.a_rows = nullptr;
    r.a_rows_n = (int)R;
    r.slab = out;
    if (int rc = launch_gemm_tn(r, (hipStream_t)stream)) return rc;
    STAR_LAUNCH_CHECK("star_gemm_tn_batched_bf16");
    return STAR_OK;
}

// Bias gradient of a projection: out[n] = sum_m x[m][n] (fp32, fixed summation order: rows of a 128-row chunk by 8
// interleaved lanes, the lanes in order, then the chunks in order -- bitwise reproducible).
namespace star {
template <typename T, int N>
struct alignas(sizeof(T) * N) Pack {
    T v[N];
};
constexpr int CS_ROWS = 128;
template <typename T>
__device__ __forceinline__ void colsum_part_body(const T* __restrict__ x, int64_t ld, int64_t M, int N,
                                                 float* __restrict__ part) {
    __shared__ float sm[8][256 + 8];
    const int tid = threadIdx.x, cg = tid & 31, rl = tid >> 5;
    const int c0 = blockIdx.x * 256 + cg * 8;
    const int64_t r0 = (int64_t)blockIdx.y * CS_ROWS;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int cc = min(c0, N - 8);                       // columns past N re-read valid ones; dropped at the store
    typedef Pack<T, 8> vec8;
    vec8 v[CS_ROWS / 8];
#pragma unroll
    for (int i = 0; i < CS_ROWS / 8; ++i) {
        const int64_t r = min(r0 + rl + 8 * i, M - 1);   // every load unconditional; rows past M are dropped below
        v[i] = *(const vec8*)(x + r * ld + cc);
    }
#pragma unroll
    for (int i = 0; i < CS_ROWS / 8; ++i) {
        const bool ok = r0 + rl + 8 * i < M;
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] += ok ? (float)v[i].v[k] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) sm[rl][cg * 8 + k] = acc[k];
    __syncthreads();
    const int c = blockIdx.x * 256 + tid;
    if (c < N) {
        float t = sm[0][tid];
#pragma unroll
        for (int q = 1; q < 8; ++q) t += sm[q][tid];
        part[(int64_t)blockIdx.y * N + c] = t;
    }
}
template <typename T>
__global__ __launch_bounds__(256) void colsum_part_kernel(const T* __restrict__ x, int64_t ld, int64_t M, int N,
                                                          float* __restrict__ part) {
    colsum_part_body<T>(x, ld, M, N, part);
}
// batched form (star_colsum_batched): matrix blockIdx.z of a list that travels in the kernel arguments
constexpr int CS_MAX_BATCH = 16;
struct ColsumList {
    const void* x[CS_MAX_BATCH];
};
template <typename T>
__global__ __launch_bounds__(256) void colsum_part_batched_kernel(ColsumList l, int64_t ld, int64_t M, int N,
                                                                  float* __restrict__ part) {
    const int bi = blockIdx.z;
    const T* x = (const T*)l.x[0];
#pragma unroll
    for (int i = 1; i < CS_MAX_BATCH; ++i)          // constant indices: no copy of the argument array to scratch
        if (bi == i) x = (const T*)l.x[i];
    colsum_part_body<T>(x, ld, M, N, part + (int64_t)bi * gridDim.y * N);
}
// (blockIdx.y: matrix of a batch; part / out advance by nchunk * N / N per matrix)
__global__ __launch_bounds__(256) void colsum_finish_kernel(const float* __restrict__ part, int nchunk, int N,
                                                            float* __restrict__ out) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= N) return;
    part += (int64_t)blockIdx.y * nchunk * N;
    out += (int64_t)blockIdx.y * N;
    float t = 0.f;
    int q = 0;
    for (; q + 3 < nchunk; q += 4) {                      // four loads in flight, the order of the adds stays fixed
        const float a0 = part[(int64_t)q * N + c], a1 = part[(int64_t)(q + 1) * N + c];
        const float a2 = part[(int64_t)(q + 2) * N + c], a3 = part[(int64_t)(q + 3) * N + c];
        t += a0;
        t += a1;
        t += a2;
        t += a3;
    }
    for (; q < nchunk; ++q) t += part[(int64_t)q * N + c];
    out[c] = t;
}

// dst_t[(z * cols + c)][r] = bf16(src_z[r][c]): the Conv1D weights of several layers ([in, out] each, fp32 or bf16),
// transposed and laid side by side as the B operand of star_gemm_nt_bf16 (rows % 64 == 0, cols % 64 == 0)
template <typename T>
__global__ __launch_bounds__(256) void transpose_list_kernel(ColsumList l, int rows, int cols, __bf16* __restrict__ dst_t) {
    __shared__ __attribute__((aligned(16))) uint8_t tile[64 * TP_PITCH];
    const int bi = blockIdx.z;
    const T* src = (const T*)l.x[0];
#pragma unroll
    for (int i = 1; i < CS_MAX_BATCH; ++i)
        if (bi == i) src = (const T*)l.x[i];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int t = threadIdx.x, c = t & 7;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const int rr = (t >> 3) + pass * 32;
        const Pack<T, 8> a = *(const Pack<T, 8>*)(src + (int64_t)(r0 + rr) * cols + c0 + c * 8);
        bf16x8 v;
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = (__bf16)a.v[k];
        *(bf16x8*)(tile + rr * TP_PITCH + c * 16) = v;
    }
    __syncthreads();
    tp_write_out(tile, dst_t + (int64_t)bi * cols * rows, rows, c0, r0);
}
}  // namespace star

extern "C" size_t star_colsum_workspace(int64_t M, int N) {
    if (M <= 0 || N <= 0) return 0;
    return (size_t)((M + star::CS_ROWS - 1) / star::CS_ROWS) * (size_t)N * sizeof(float);
}

extern "C" int star_colsum(const void* x, int dtype, int64_t M, int N, int64_t ld, float* out, void* workspace,
                           size_t workspace_bytes, void* stream) {
    using namespace star;
    STAR_REQUIRE(x && out && workspace, "star_colsum: null pointer");
    STAR_REQUIRE(dtype == STAR_DT_BF16 || dtype == STAR_DT_F32, "star_colsum: bad dtype %d", dtype);
    STAR_REQUIRE(M >= 1 && N >= 8 && N % 8 == 0 && ld >= N && ld % 8 == 0, "star_colsum: bad shape M=%lld N=%d ld=%lld",
                 (long long)M, N, (long long)ld);
    STAR_REQUIRE(((uintptr_t)x & (dtype == STAR_DT_BF16 ? 15 : 31)) == 0, "star_colsum: x must be aligned to 8 elements");
    const int64_t nchunk = (M + CS_ROWS - 1) / CS_ROWS;
    STAR_REQUIRE(nchunk <= 65535, "star_colsum: M=%lld too large", (long long)M);
    if (workspace_bytes < star_colsum_workspace(M, N)) {
        set_error("star_colsum: workspace %zu < %zu bytes", workspace_bytes, star_colsum_workspace(M, N));
        return STAR_ERR_WORKSPACE;
    }
    const dim3 grid(cdiv(N, 256), (unsigned)nchunk);
    if (dtype == STAR_DT_BF16)
        hipLaunchKernelGGL(colsum_part_kernel<__bf16>, grid, dim3(256), 0, (hipStream_t)stream, (const __bf16*)x, ld, M, N,
                           (float*)workspace);
    else
        hipLaunchKernelGGL(colsum_part_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)x, ld, M, N,
                           (float*)workspace);
    hipLaunchKernelGGL(colsum_finish_kernel, dim3(cdiv(N, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const float*)workspace, (int)nchunk, N, out);
    STAR_LAUNCH_CHECK("star_colsum");
    return STAR_OK;
}

extern "C" int star_colsum_batched(const void* const* x_list, int nbatch, int dtype, int64_t M, int N, int64_t ld,
                                   float* out, void* workspace, size_t workspace_bytes, void* stream) {
    using namespace star;
    STAR_REQUIRE(x_list && out && workspace, "star_colsum_batched: null pointer");
    STAR_REQUIRE(nbatch >= 1 && nbatch <= CS_MAX_BATCH, "star_colsum_batched: nbatch = %d (1..%d)", nbatch, CS_MAX_BATCH);
    STAR_REQUIRE(dtype == STAR_DT_BF16 || dtype == STAR_DT_F32, "star_colsum_batched: bad dtype %d", dtype);
    STAR_REQUIRE(M >= 1 && N >= 8 && N % 8 == 0 && ld >= N && ld % 8 == 0,
                 "star_colsum_batched: bad shape M=%lld N=%d ld=%lld", (long long)M, N, (long long)ld);
    ColsumList l = {};
    for (int i = 0; i < nbatch; ++i) {
        STAR_REQUIRE(x_list[i] && ((uintptr_t)x_list[i] & (dtype == STAR_DT_BF16 ? 15 : 31)) == 0,
                     "star_colsum_batched: matrix %d is null or not aligned to 8 elements", i);
        l.x[i] = x_list[i];
    }
    const int64_t nchunk = (M + CS_ROWS - 1) / CS_ROWS;
    STAR_REQUIRE(nchunk <= 65535, "star_colsum_batched: M=%lld too large", (long long)M);
    if (workspace_bytes < (size_t)nbatch * star_colsum_workspace(M, N)) {
        set_error("star_colsum_batched: workspace %zu < %zu bytes", workspace_bytes,
                  (size_t)nbatch * star_colsum_workspace(M, N));
        return STAR_ERR_WORKSPACE;
    }
    const dim3 grid(cdiv(N, 256), (unsigned)nchunk, (unsigned)nbatch);
    if (dtype == STAR_DT_BF16)
        hipLaunchKernelGGL(colsum_part_batched_kernel<__bf16>, grid, dim3(256), 0, (hipStream_t)stream, l, ld, M, N,
                           (float*)workspace);
    else
        hipLaunchKernelGGL(colsum_part_batched_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, l, ld, M, N,
                           (float*)workspace);
    hipLaunchKernelGGL(colsum_finish_kernel, dim3(cdiv(N, 256), nbatch), dim3(256), 0, (hipStream_t)stream,
                       (const float*)workspace, (int)nchunk, N, out);
    STAR_LAUNCH_CHECK("star_colsum_batched");
    return STAR_OK;
}

extern "C" int star_transpose_list_bf16(const void* const* src_list, int n, int dtype, int rows, int cols, void* dst_t,
                                        void* stream) {
    using namespace star;
    STAR_REQUIRE(src_list && dst_t, "star_transpose_list_bf16: null pointer");
    STAR_REQUIRE(n >= 1 && n <= CS_MAX_BATCH, "star_transpose_list_bf16: n = %d (1..%d)", n, CS_MAX_BATCH);
    STAR_REQUIRE(dtype == STAR_DT_BF16 || dtype == STAR_DT_F32, "star_transpose_list_bf16: bad dtype %d", dtype);
    STAR_REQUIRE(rows >= 64 && cols >= 64 && rows % 64 == 0 && cols % 64 == 0,
                 "star_transpose_list_bf16: rows = %d, cols = %d must be multiples of 64", rows, cols);
    STAR_REQUIRE(((uintptr_t)dst_t & 15) == 0, "star_transpose_list_bf16: dst_t must be 16-byte aligned");
    ColsumList l = {};
    for (int i = 0; i < n; ++i) {
        STAR_REQUIRE(src_list[i] && ((uintptr_t)src_list[i] & (dtype == STAR_DT_BF16 ? 15 : 31)) == 0,
                     "star_transpose_list_bf16: matrix %d is null or not aligned to 8 elements", i);
        l.x[i] = src_list[i];
    }
    const dim3 grid(cols / 64, rows / 64, n);
    if (dtype == STAR_DT_BF16)
        hipLaunchKernelGGL(transpose_list_kernel<__bf16>, grid, dim3(256), 0, (hipStream_t)stream, l, rows, cols,
                           (__bf16*)dst_t);
    else
        hipLaunchKernelGGL(transpose_list_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, l, rows, cols,
                           (__bf16*)dst_t);
    STAR_LAUNCH_CHECK("star_transpose_list_bf16");
    return STAR_OK;
}

extern "C" int star_cast_transpose_bf16(const float* src, int rows, int cols, void* dst, void* dst_t,
                                        const float* t_scale, void* stream) {
    using namespace star;
    STAR_REQUIRE(src && (dst || dst_t), "star_cast_transpose_bf16: null pointer");
    STAR_REQUIRE(rows >= 1 && cols >= 1, "star_cast_transpose_bf16: bad shape %d x %d", rows, cols);
    const bool fast = rows % 64 == 0 && cols % 64 == 0 && ((uintptr_t)src % 16 == 0) && ((uintptr_t)dst % 16 == 0) &&
                      ((uintptr_t)dst_t % 16 == 0);
    if (fast)
        hipLaunchKernelGGL(cast_transpose_kernel, dim3(cols / 64, rows / 64), dim3(256), 0, (hipStream_t)stream, src,
                           rows, cols, (__bf16*)dst, (__bf16*)dst_t, t_scale);
    else
        hipLaunchKernelGGL(cast_transpose_generic_kernel, dim3(cdiv(cols, 64), cdiv(rows, 64)), dim3(256), 0,
                           (hipStream_t)stream, src, rows, cols, (__bf16*)dst, (__bf16*)dst_t, t_scale);
    STAR_LAUNCH_CHECK("star_cast_transpose_bf16");
    return STAR_OK;
}
