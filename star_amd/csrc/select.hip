// Top-k anchor selection, fused with the nugget type table (SURVEY.md section 8 rows a2 + a3).
//
// Reference behaviour restated (src/models/w2v_asr.py:208-244, 316-323): per utterance, rank the T
// frames by (key descending, frame index ascending) where key = score, pads = finfo.min, last valid
// frame = finfo.max; the first k_b ranks are kept, ranks k_b..K-1 fill the pad slots; both groups
// are emitted in ascending frame order.  The reference does this with two full argsorts, a scatter,
// two gathers and an index_put (~12 launches); here one workgroup per utterance does
//   keys -> LDS, two 4-pass 8-bit radix selects (k_b-th and K-th largest), tie ranks by frame
//   order, and an order-preserving compaction (block prefix sums) -- integer exact, no sort.
// HBM traffic: reads 4*T (+T mask) bytes, writes 8*K idx + K mask + 4*K scores + 8*K pos + 8*T table.
#include "common.h"

namespace star {

constexpr int SEL_THREADS = 1024;
constexpr int SEL_WAVES = SEL_THREADS / 64;
static_assert(SEL_WAVES <= 16, "the per-wave totals live in misc[16]");

__device__ __forceinline__ uint32_t order_key(float f) {
    // unsigned order of the result == float order; -0.0 == +0.0; NaN ranks above everything
    if (f != f) return 0xFFFFFFFFu;
    uint32_t u = __float_as_uint(f);
    if (u == 0x80000000u) u = 0u;
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// exclusive block scan of a packed pair of 16-bit counters; `tot` receives the block total
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t* wtot, uint32_t& tot) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    uint32_t incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        uint32_t n = __shfl_up(incl, o, 64);
        if (lane >= o) incl += n;
    }
    if (lane == 63) wtot[wid] = incl;
    __syncthreads();
    uint32_t base = 0, all = 0;
#pragma unroll
    for (int w = 0; w < SEL_WAVES; ++w) {
        uint32_t t = wtot[w];
        if (w < wid) base += t;
        all += t;
    }
    __syncthreads();
    tot = all;
    return base + incl - v;
}

// k-th largest key (k >= 1, k <= T): returns the threshold key `v` and `need` = how many of the
// keys equal to v belong to the top-k (the rest of the top-k are the keys > v).
__device__ void radix_select(const uint32_t* keys, int T, uint32_t k, uint32_t* hist, uint32_t* bc,
                             uint32_t& v, uint32_t& need) {
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    uint32_t prefix = 0, pmask = 0, remaining = k;
    for (int shift = 24; shift >= 0; shift -= 8) {
        if (tid < 256) hist[tid] = 0;  // 256 bins
        __syncthreads();
        for (int i = tid; i < T; i += SEL_THREADS) {
            uint32_t key = keys[i];
            if ((key & pmask) == prefix) atomicAdd(&hist[(key >> shift) & 255u], 1u);
        }
        __syncthreads();
        if (wid == 0) {
            // lane l owns bins 255-4l .. 252-4l, i.e. lanes ascend while digits descend
            uint32_t c[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) c[j] = hist[255 - 4 * lane - j];
            uint32_t s = c[0] + c[1] + c[2] + c[3];
            uint32_t incl = s;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                uint32_t n = __shfl_up(incl, o, 64);
                if (lane >= o) incl += n;
            }
            uint32_t run = incl - s;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (run < remaining && remaining <= run + c[j]) {
                    bc[0] = (uint32_t)(255 - 4 * lane - j);
                    bc[1] = remaining - run;
                }
                run += c[j];
            }
        }
        __syncthreads();
        prefix |= bc[0] << shift;
        pmask |= 0xFFu << shift;
        remaining = bc[1];
        __syncthreads();
    }
    v = prefix;
    need = remaining;
}

__global__ __launch_bounds__(SEL_THREADS) void topk_select_kernel(
    const float* __restrict__ scores, const uint8_t* __restrict__ attn_mask,
    const int32_t* __restrict__ lengths, const int32_t* __restrict__ n_nugget, float ratio, int T,
    int K, int64_t* __restrict__ idx_out, uint8_t* __restrict__ mask_out, int32_t* __restrict__ flat_rows,
    float* __restrict__ scores_gather, int64_t* __restrict__ type_table,
    int64_t* __restrict__ positions) {
    extern __shared__ uint32_t smem[];
    uint32_t* keys = smem;                        // [T]
    uint32_t* hist = keys + T;                    // [256]
    uint32_t* misc = hist + 256;                  // [16]
    float* sraw = (float*)(misc + 16);            // [T] the scores as read (gathered again at the end)
    uint8_t* cls = (uint8_t*)(sraw + T);          // [T]
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int b = blockIdx.x;
    const float* srow = scores + (size_t)b * T;
    const uint8_t* mrow = attn_mask ? attn_mask + (size_t)b * T : nullptr;

    // ---- number of valid frames
    int n_tok;
    if (mrow) {
        int c = 0;
        for (int i = tid; i < T; i += SEL_THREADS) c += (mrow[i] != 0);
        c = wave_sum(c);
        if (lane == 0) misc[wid] = (uint32_t)c;
        __syncthreads();
        n_tok = 0;
#pragma unroll
        for (int w = 0; w < SEL_WAVES; ++w) n_tok += (int)misc[w];
        __syncthreads();
    } else {
        n_tok = lengths[b];
        if (n_tok > T) n_tok = T;
    }
    int kb;
    if (n_nugget) {
        kb = n_nugget[b];
    } else {
        // one rounding per op, exactly as torch evaluates  n_tok * ratio + 0.99  in fp32
        float kf = __fadd_rn(__fmul_rn((float)n_tok, ratio), 0.99f);
        kb = (int)kf;
        if (kb > n_tok) kb = n_tok;
    }
    if (kb > K) kb = K;
    if (kb < 0) kb = 0;

    // ---- selection keys
    const uint32_t key_pad = order_key(-3.4028234663852886e38f);
    const uint32_t key_last = order_key(3.4028234663852886e38f);
    for (int i = tid; i < T; i += SEL_THREADS) {
        bool valid = mrow ? (mrow[i] != 0) : (i < n_tok);
        const float sv = srow[i];
        sraw[i] = sv;
        uint32_t key = valid ? order_key(sv) : key_pad;
        if (i == n_tok - 1) key = key_last;
        keys[i] = key;
    }
    __syncthreads();

    uint32_t v1 = 0xFFFFFFFFu, need1 = 0, v2, need2;
    if (kb > 0) radix_select(keys, T, (uint32_t)kb, hist, misc, v1, need1);
    if (kb == K) {           // a full-length utterance: the top-K set is the kept set (block-uniform branch)
        v2 = v1;
        need2 = need1;
    } else {
        radix_select(keys, T, (uint32_t)K, hist, misc, v2, need2);
    }

    // ---- tie ranks in frame order: each thread owns a contiguous chunk of frames
    const int C = (T + SEL_THREADS - 1) / SEL_THREADS;
    const int lo = min(tid * C, T), hi = min(lo + C, T);
    uint32_t e = 0;
    for (int i = lo; i < hi; ++i) {
        uint32_t key = keys[i];
        e += (key == v1) + ((uint32_t)(key == v2) << 16);
    }
    uint32_t tot;
    uint32_t ebase = block_excl_scan(e, misc, tot);
    uint32_t r1 = ebase & 0xFFFFu, r2 = ebase >> 16;
    uint32_t cnt = 0;
    for (int i = lo; i < hi; ++i) {
        uint32_t key = keys[i];
        bool kept = (kb > 0) && (key > v1 || (key == v1 && r1 < need1));
        bool intop = key > v2 || (key == v2 && r2 < need2);
        r1 += (key == v1);
        r2 += (key == v2);
        uint8_t c = kept ? 1 : (intop ? 2 : 0);
        cls[i] = c;
        cnt += (c == 1) + ((uint32_t)(c == 2) << 16);
    }
    uint32_t pbase = block_excl_scan(cnt, misc, tot);
    uint32_t p1 = pbase & 0xFFFFu, p2 = pbase >> 16;
    int64_t* irow = idx_out + (size_t)b * K;
    for (int i = lo; i < hi; ++i) {
        uint8_t c = cls[i];
        if (c == 0) continue;
        uint32_t slot = (c == 1) ? p1++ : (uint32_t)kb + p2++;
        if (slot < (uint32_t)K) {
            irow[slot] = i;
            if (flat_rows) flat_rows[(size_t)b * K + slot] = b * T + i;
            if (scores_gather) scores_gather[(size_t)b * K + slot] = sraw[i];
            if (positions) positions[(size_t)b * K + slot] = (c == 1) ? (int64_t)i : (int64_t)-1;
        }
    }
    if (mask_out)
        for (int j = tid; j < K; j += SEL_THREADS) mask_out[(size_t)b * K + j] = (uint8_t)(j < kb);
    __syncthreads();
    if (type_table)
        for (int i = tid; i < T; i += SEL_THREADS) type_table[(size_t)b * T + i] = (cls[i] == 1) ? 1 : 0;
}

// stand-alone add_nugget_type_embedding (w2v_asr.py:316-323) for callers that already hold indices + mask
__global__ __launch_bounds__(SEL_THREADS) void type_table_kernel(const int64_t* __restrict__ idx,
                                                                 const uint8_t* __restrict__ nmask, int T, int K,
                                                                 int64_t* __restrict__ table,
                                                                 int64_t* __restrict__ positions) {
    const int b = blockIdx.x, tid = threadIdx.x;
    for (int i = tid; i < T; i += SEL_THREADS) table[(size_t)b * T + i] = 0;
    __syncthreads();
    for (int j = tid; j < K; j += SEL_THREADS) {
        const bool keep = nmask[(size_t)b * K + j] != 0;
        const int64_t t = idx[(size_t)b * K + j];
        if (positions) positions[(size_t)b * K + j] = keep ? t : (int64_t)-1;
        if (keep && t >= 0 && t < T) table[(size_t)b * T + t] = 1;
    }
}

}  // namespace star

extern "C" int star_nugget_type_table(const int64_t* idx, const uint8_t* nugget_mask, int B, int T, int K,
                                      int64_t* type_table, int64_t* positions, void* stream) {
    using namespace star;
    STAR_REQUIRE(idx && nugget_mask && type_table, "star_nugget_type_table: null pointer");
    STAR_REQUIRE(B >= 1 && T >= 1 && K >= 1, "star_nugget_type_table: bad shape B=%d T=%d K=%d", B, T, K);
    hipLaunchKernelGGL(type_table_kernel, dim3(B), dim3(SEL_THREADS), 0, (hipStream_t)stream, idx, nugget_mask, T, K,
                       type_table, positions);
    STAR_LAUNCH_CHECK("star_nugget_type_table");
    return STAR_OK;
}

extern "C" int star_topk_select(const float* scores, const uint8_t* attn_mask, const int32_t* lengths,
                                const int32_t* n_nugget, float ratio, int B, int T, int K,
                                int64_t* idx_out, uint8_t* mask_out, float* scores_gather,
                                int64_t* type_table, int64_t* positions, int32_t* flat_rows, void* stream) {
    using namespace star;
    STAR_REQUIRE(scores && idx_out, "star_topk_select: scores and idx_out must be non-null");
    STAR_REQUIRE(!flat_rows || (int64_t)B * T < (1ll << 31), "star_topk_select: B*T too large for int32 flat rows");
    STAR_REQUIRE((attn_mask != nullptr) != (lengths != nullptr),
                 "star_topk_select: give exactly one of attn_mask / lengths");
    STAR_REQUIRE(B >= 1 && T >= 1 && K >= 1 && K <= T, "star_topk_select: need B>=1, 1<=K<=T (B=%d T=%d K=%d)", B, T, K);
    STAR_REQUIRE(T <= 16384, "star_topk_select: T=%d exceeds the 16384-frame LDS row limit", T);
    size_t smem = (size_t)T * 8 + 256 * 4 + 16 * 4 + (size_t)T;
    if (smem > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)topk_select_kernel,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) {
            set_error("star_topk_select: cannot reserve %zu bytes of LDS: %s", smem, hipGetErrorString(e));
            return STAR_ERR_LAUNCH;
        }
    }
    hipLaunchKernelGGL(topk_select_kernel, dim3(B), dim3(SEL_THREADS), smem, (hipStream_t)stream, scores,
                       attn_mask, lengths, n_nugget, ratio, T, K, idx_out, mask_out, flat_rows, scores_gather,
                       type_table, positions);
    STAR_LAUNCH_CHECK("star_topk_select");
    return STAR_OK;
}
