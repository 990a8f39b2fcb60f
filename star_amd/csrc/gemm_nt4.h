// Scorer forward GEMM, 4-wave form (included by gemm.hip, which owns GemmParams).
//
// Same 256 x 256 output tile and the same EPI_SCORE results layout as gemm_nt_kernel<EPI_SCORE> (partial scores per column
// tile, ReLU sign bits in relu_bit_dword / relu_bit_pos order), different machine shape:
//   * 256 threads = 4 waves (2 x 2), ONE per SIMD, each 128 x 128 = 8 x 8 mfma_f32_16x16x32_bf16 accumulators (256
//     registers, the AGPR half of the 512-register budget).  LDS fragment bytes per FLOP are 2/3 of the 8-wave kernel's
//     and no two waves share a SIMD, so nothing has to be handed over between them: one barrier per 32 k (64 MFMAs per
//     wave) instead of eight per 64 k.
//   * operands travel HBM/L2 -> registers -> LDS (global_load_dwordx4, ds_write_b128): a wave that is alone on its SIMD
//     cannot afford the issue cost of LDS-DMA pieces between its MFMAs (60-185 cycles each, MI355X_MICROARCH.md).  A (the
//     frames, from HBM): four register sets of 4 x 16 B per lane -- the loads of K stage t + 6 are issued during stage t,
//     written to the LDS ring (2 stages of 32 k: A part 256 rows x 64 B | B part) during stage t + 4, read as fragments
//     during stage t + 5 and multiplied in stage t + 6.  B (W1, L2-resident): two sets, loaded two stages later.
//   * fragments are single-buffered and refilled in place: even stages run row-major over the 8 x 8 accumulator tiles
//     (A fragment i is dead after row i and takes stage t + 1's rows at once; the B fragments retire along the last
//     row), odd stages column-major (the mirror image), so every fragment is re-read at least 8 MFMAs before its next use.
//   * persistent: a workgroup walks tiles vb, vb + gridDim.x, ...; the K stages of the next tile follow the last stages
//     of the current one without a pipeline restart, and the epilogue runs with those loads in flight.
// LDS rows are 64 B (4 chunks of 16 B); chunk c of row r sits at position c ^ swz4(r), swz4 = [0,2,3,1][(r >> 2) & 3]:
// the 16 lanes one ds_read_b128 cycle serves (rows 0-3 and 12-15 of chunk q with rows 4-11 of chunk q + 1) then fall on
// 16 distinct 16-B slots of the 256-B bank row; the staging stores (8 contiguous lanes = 2 rows) are conflict-free anyway.
#pragma once
#include <type_traits>

namespace star {

constexpr int G4_THREADS = 256;
constexpr int G4_BK = 32;
constexpr int G4_ASETS = 2;                      // A staging depth in register sets (2 or 4)
constexpr int G4_PART = 256 * 64;                 // one operand of one stage: 256 rows x 64 B
constexpr int G4_STAGE = 2 * G4_PART;             // A part | B part
constexpr int G4_RING = 2 * G4_STAGE;             // 64 KiB
constexpr int G4_RED_OFF = G4_RING;               // float red[256 rows][2 wave columns]
constexpr int G4_BW_OFF = G4_RED_OFF + 2048;      // b1 | w2 of all columns (d <= SCORE_BW_MAXN)
constexpr int G4_LDS = G4_BW_OFF + 2 * SCORE_BW_MAXN * 4;

__device__ __forceinline__ int swz4(int row) { return (0x78 >> (((row >> 2) & 3) * 2)) & 3; }

// acc += a . b with the accumulator pinned to the AGPR file and updated in place: the 64 accumulator tiles are exactly the
// 256 AGPRs, and with the builtin the register allocator (no slack in that file) rotates tiles through VGPRs and spills
__device__ __forceinline__ void g4_mfma(f32x4& c, const bf16x8& a, const bf16x8& b) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}

// the first product of an output tile: acc = a . b (no zero-fill of 256 accumulator registers per tile)
__device__ __forceinline__ void g4_mfma0(f32x4& c, const bf16x8& a, const bf16x8& b) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=a"(c) : "v"(a), "v"(b));
}

__device__ __forceinline__ void g4_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <bool BW_STAGED, bool LIVE, bool BITS>
__global__ __launch_bounds__(G4_THREADS) __attribute__((amdgpu_waves_per_eu(1, 1))) void scorer_fwd4_kernel(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 1, wc = wid & 1;
    const int nbm = (p.M + BM - 1) / BM, nbn = (p.N + BN - 1) / BN;
    const int nk = p.kchunks * 2;                 // K stages of 32 per tile: a multiple of 4, >= 8 (the launcher checks)

    // XCD-aware persistent walk: blocks b and b + 8 share an XCD; the nbn column tiles of a row panel are neighbours on it
    auto map_tile = [&](int vb, int& mt, int& nt) -> bool {
        const int xcd = vb & 7, local = vb >> 3;
        const int panel = (local / nbn) * 8 + xcd;
        nt = local % nbn;
        if constexpr (LIVE) {
            if (panel >= p.n_live) return false;
            mt = p.live_tiles[panel];
            return true;
        }
        mt = panel;
        return panel < nbm;
    };
    int vb = blockIdx.x, m_tile, n_tile;
    if (!map_tile(vb, m_tile, n_tile)) return;

    // staging piece q (0-3: A, 4-7: B) of this wave: rows ((q & 3) * 4 + wid) * 16 + (lane >> 2) of the part, LDS
    // position lane & 3 of the row, source chunk (lane & 3) ^ swz4(row)
    const int prow = lane >> 2;
    auto a_row_of = [&](int mt, int q) {
        int am = mt * BM + (q * 4 + wid) * 16 + prow;
        return am < p.M ? am : p.M - 1;
    };
    // byte offsets from p.A / p.B (32 bits: the launcher sends operands of 4 GiB or more to the 8-wave kernel); the K
    // position is uniform and lives in scalar registers (ka, kb)
    auto a_off = [&](int64_t row, int q) {
        const int r = (q * 4 + wid) * 16 + prow;
        return (uint32_t)((row * p.lda + ((lane & 3) ^ swz4(r)) * 8) * 2);
    };
    auto b_off = [&](int nt, int q) {
        const int r = (q * 4 + wid) * 16 + prow;
        int bn = nt * BN + r;
        if (bn >= p.N) bn = p.N - 1;
        return (uint32_t)(((int64_t)bn * p.ldb + ((lane & 3) ^ swz4(r)) * 8) * 2);
    };
    uint32_t ao[4], bo[4];
    int ka = 0, kb = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        int64_t row = a_row_of(m_tile, q);
        if (p.a_rows) row = p.a_rows[row < p.a_rows_n ? row : 0];
        ao[q] = a_off(row, q);
        bo[q] = b_off(n_tile, q);
    }
    const char* const Ab = (const char*)p.A;
    const char* const Bb = (const char*)p.B;
    const int w_off = wid * 1024 + lane * 16;                    // + part * G4_PART + (q & 3) * 4096
    const int frow = lane & 15;
    const int f_off = frow * 64 + (((lane >> 4) ^ swz4(frow)) << 4);
    const int a_rd = wr * 8192 + f_off;                          // + i * 1024
    const int b_rd = G4_PART + wc * 8192 + f_off;                // + j * 1024

    u32x4 sa[G4_ASETS][4], sb[2][4];       // staging sets: A four deep (HBM latency), B two deep (W1 lives in L2)
    bf16x8 af[8], bfr[8];
    f32x4 acc[8][8];

    auto load_a = [&](auto SET_) {
        constexpr int SET = decltype(SET_)::value;
#pragma unroll
        for (int q = 0; q < 4; ++q) sa[SET][q] = *(const u32x4*)(Ab + ka + ao[q]);
        ka += G4_BK * 2;
    };
    auto load_b = [&](auto SET_) {
        constexpr int SET = decltype(SET_)::value;
#pragma unroll
        for (int q = 0; q < 4; ++q) sb[SET][q] = *(const u32x4*)(Bb + kb + bo[q]);
        kb += G4_BK * 2;
    };
    auto write_a = [&](auto SET_, int slot) {
        constexpr int SET = decltype(SET_)::value;
#pragma unroll
        for (int q = 0; q < 4; ++q) *(u32x4*)(lds + slot * G4_STAGE + q * 4096 + w_off) = sa[SET][q];
    };
    auto write_b = [&](auto SET_, int slot) {
        constexpr int SET = decltype(SET_)::value;
#pragma unroll
        for (int q = 0; q < 4; ++q) *(u32x4*)(lds + slot * G4_STAGE + G4_PART + q * 4096 + w_off) = sb[SET][q];
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>;

    // ---- prologue (first tile of the workgroup).  Invariant at the top of stage t: fragments = stage t, ring slot
    // (t + 1) & 1 = stage t + 1, A sets (t + 2 .. t + 5) & 3 = stages t + 2 .. t + 5, B sets t & 1, (t + 1) & 1 = stages
    // t + 2, t + 3
    load_a(I0{});
    load_b(I0{});
    load_a(I1{});
    load_b(I1{});
    if constexpr (G4_ASETS == 4) {
        load_a(I2{});
        load_a(I3{});
    }
    float* bw_lds = (float*)(lds + G4_BW_OFF);
    if constexpr (BW_STAGED) {
        for (int n = tid; n < nbn * BN; n += G4_THREADS) {
            bw_lds[n] = n < p.N ? p.b1[n] : 0.f;
            bw_lds[SCORE_BW_MAXN + n] = n < p.N ? p.w2[n] : 0.f;
        }
    }
    write_a(I0{}, 0);
    write_b(I0{}, 0);
    load_b(I0{});
    load_a(I0{});
    write_a(I1{}, 1);
    write_b(I1{}, 1);
    load_b(I1{});
    load_a(I1{});
    g4_barrier();
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        af[i] = *(const bf16x8*)(lds + a_rd + i * 1024);
        bfr[i] = *(const bf16x8*)(lds + b_rd + i * 1024);
    }
    g4_barrier();       // stage 0 starts writing ring slot 0: every wave has its fragments of it

    // one K stage: 64 MFMAs on the fragments in registers; refill them with stage t + 1 (ring slot (S + 1) & 1); move A set
    // (S + 2) & 3 and B set S & 1 (stage t + 2) to ring slot S & 1 and reload them with stages t + 6 (A) and t + 4 (B).  The
    // B loads go first: a load's data waits for every older load, and the B pieces are needed two stages earlier
    auto stage = [&](auto S_, auto FIRST_) {
        constexpr int S = decltype(S_)::value;
        constexpr bool FIRST = decltype(FIRST_)::value;      // first stage of an output tile: the products start the sums
        constexpr int SETA = G4_ASETS == 4 ? (S + 2) & 3 : S & 1, SETB = S & 1;
        const uint8_t* rd = lds + ((S + 1) & 1) * G4_STAGE;
        uint8_t* wp = lds + (S & 1) * G4_STAGE + w_off;
#pragma unroll
        for (int o = 0; o < 8; ++o) {
#pragma unroll
            for (int in = 0; in < 8; ++in) {
                if constexpr ((S & 1) == 0) {
                    if constexpr (FIRST) g4_mfma0(acc[o][in], af[o], bfr[in]);
                    else g4_mfma(acc[o][in], af[o], bfr[in]);
                    if (o == 7 && !(STAR_GEMM_ABL & 2)) bfr[in] = *(const bf16x8*)(rd + b_rd + in * 1024);
                } else {
                    g4_mfma(acc[in][o], af[in], bfr[o]);
                    if (o == 7 && !(STAR_GEMM_ABL & 2)) af[in] = *(const bf16x8*)(rd + a_rd + in * 1024);
                }
            }
            if constexpr (!(STAR_GEMM_ABL & 2)) {
                if constexpr ((S & 1) == 0) af[o] = *(const bf16x8*)(rd + a_rd + o * 1024);
                else bfr[o] = *(const bf16x8*)(rd + b_rd + o * 1024);
            }
            if (o < 4) {
                if constexpr (!(STAR_GEMM_ABL & 32)) *(u32x4*)(wp + G4_PART + o * 4096) = sb[SETB][o];
                if constexpr (!(STAR_GEMM_ABL & 1)) sb[SETB][o] = *(const u32x4*)(Bb + kb + bo[o]);
            } else {
                if constexpr (!(STAR_GEMM_ABL & 32)) *(u32x4*)(wp + (o - 4) * 4096) = sa[SETA][o - 4];
                if constexpr (!(STAR_GEMM_ABL & 1)) sa[SETA][o - 4] = *(const u32x4*)(Ab + ka + ao[o - 4]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        ka += G4_BK * 2;
        kb += G4_BK * 2;
        g4_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };

    for (;;) {      // output tiles of this workgroup
        int nm_tile = m_tile, nn_tile = n_tile;
        const bool has_next = map_tile(vb + (int)gridDim.x, nm_tile, nn_tile);
        if (!has_next) {        // the look-ahead loads of the last tile re-read its own first stages (never used)
            nm_tile = m_tile;
            nn_tile = n_tile;
        }
        // row ids of the next tile's A pieces: loaded now, used when the stream switches over (8 stages before the end)
        int32_t nrow[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = a_row_of(nm_tile, q);
            nrow[q] = p.a_rows ? p.a_rows[r < p.a_rows_n ? r : 0] : r;
        }
        auto group = [&](int t, auto FIRST_) {       // four K stages; t = index of the first
            if (t == nk - 4) {      // the B loads of stage t + 0 + 4 = nk are the next tile's stage 0 (A: with 2 sets)
#pragma unroll
                for (int q = 0; q < 4; ++q) bo[q] = b_off(nn_tile, q);
                kb = 0;
                if constexpr (G4_ASETS == 2) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) ao[q] = a_off(nrow[q], q);
                    ka = 0;
                }
            }
            stage(I0{}, FIRST_);
            stage(I1{}, std::false_type{});
            if constexpr (G4_ASETS == 4) {
                if (t == nk - 8) {      // the A loads of stage t + 2 + 6 = nk likewise
#pragma unroll
                    for (int q = 0; q < 4; ++q) ao[q] = a_off(nrow[q], q);
                    ka = 0;
                }
            }
            stage(I2{}, std::false_type{});
            stage(I3{}, std::false_type{});
        };
        group(0, std::true_type{});
        for (int t = 4; t < nk; t += 4) group(t, std::false_type{});

        // (the MFMAs are inline asm, invisible to the compiler's hazard recogniser: keep the last of them clear of the first
        // accumulator read by more than its four passes)
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
        // ---- epilogue: ReLU / bias / w2 row dot -> partial scores of this column tile; ReLU sign bits in the layout of
        // relu_bit_dword / relu_bit_pos (this wave's 128 columns are the 8-wave kernel's wave columns 2 wc and 2 wc + 1)
        // (the lane id goes through an empty asm so that nothing of the epilogue's addressing is hoisted above the K loop,
        // where every VGPR is spoken for)
        int lane_e = lane, tid_e = tid;
        asm volatile("" : "+v"(lane_e), "+v"(tid_e));
        const int ccol = lane_e & 15, crow4 = (lane_e >> 4) * 4;
        float* red = (float*)(lds + G4_RED_OFF);
        // this wave's b1 / w2 (column n0 + wc * 128 + tn * 16 + (lane_e & 15)); the global form waits behind the look-ahead loads
        float bb[8], ww[8];
#pragma unroll
        for (int tn = 0; tn < 8; ++tn) {
            const int n = n_tile * BN + wc * 128 + tn * 16 + (lane_e & 15);
            if constexpr (BW_STAGED) {
                bb[tn] = bw_lds[n];
                ww[tn] = bw_lds[SCORE_BW_MAXN + n];
            } else {
                bb[tn] = n < p.N ? p.b1[n] : 0.f;
                ww[tn] = n < p.N ? p.w2[n] : 0.f;
            }
        }
        uint32_t hb[2][4] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}};
#pragma unroll
        for (int tm = 0; tm < 8; ++tm) {
            float srow[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int tn = 0; tn < 8; ++tn)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float h = acc[tm][tn][j] + bb[tn];
                    srow[j] += fmaxf(h, 0.f) * ww[tn];
                    if constexpr (BITS)
                        hb[tn >> 2][tm >> 1] =
                            __builtin_amdgcn_alignbit(hb[tn >> 2][tm >> 1], __float_as_uint(0.f - h), 31);
                }
#define STAR_ROR_ADD4(CTRL)                                                                                  \
    _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                            \
        srow[j] += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(srow[j]), (CTRL), 0xF, 0xF, false))
            STAR_ROR_ADD4(0x128);
            STAR_ROR_ADD4(0x124);
            STAR_ROR_ADD4(0x122);
            STAR_ROR_ADD4(0x121);
#undef STAR_ROR_ADD4
            const float v01 = ccol == 0 ? srow[0] : srow[1], v23 = ccol == 2 ? srow[2] : srow[3];
            if (ccol < 4) red[(wr * 128 + tm * 16 + crow4 + ccol) * 2 + wc] = ccol < 2 ? v01 : v23;
            // one row of accumulator tiles at a time: the staging sets and fragments of the next tile stay in registers
            // through the epilogue, and a scheduler that reads many tiles ahead pushes them to scratch -- inside the K loop
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (BITS) {       // (a template parameter: under a run-time test hipcc sinks the bit packing into a
                                    // second pass over the accumulators and keeps all 256 h values in VGPRs for it)
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                uint32_t* dst = p.relu_bits +
                                ((int64_t)(m_tile * nbn + n_tile) * 8 + wr * 4 + wc * 2 + half) * (4 * 64) + lane_e;
#pragma unroll
                for (int q = 0; q < 4; ++q) dst[q * 64] = hb[half][q];
            }
        }
        g4_barrier();
        {
            const float r0 = red[tid_e * 2], r1 = red[tid_e * 2 + 1];
            p.partial[(int64_t)n_tile * p.partial_ld + m_tile * BM + tid_e] = r0 + r1;
        }
        // (the next write to `red` is a whole K loop of barriers away)
        if (!has_next) break;
        vb += (int)gridDim.x;
        m_tile = nm_tile;
        n_tile = nn_tile;
        // the next tile's stage-0 fragments again, from ring slot 0 (the last stage read them already, but carrying 64
        // fragment registers through the epilogue costs more than these 16 reads); the barrier keeps stage 0's ring
        // writes of a fast wave behind the reads of a slow one
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            af[i] = *(const bf16x8*)(lds + a_rd + i * 1024);
            bfr[i] = *(const bf16x8*)(lds + b_rd + i * 1024);
        }
        g4_barrier();
    }
    // retire the look-ahead loads before the wave ends
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

}  // namespace star
