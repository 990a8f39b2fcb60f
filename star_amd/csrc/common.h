// Shared helpers for the libstar_nugget.so kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <atomic>
#include "../../include/star_nugget.h"

namespace star {

void set_error(const char* fmt, ...);
// test-only kernel-variant switches (star_test_override in star_nugget.h); 0 = the library's own choice
int test_override(int which);

// "done once per device" flag for idempotent set-up calls (hipFuncSetAttribute costs tens of microseconds on the
// host): a bit per device, set with release order after the call succeeded.  Two threads that race on the first
// use both make the (idempotent, thread-safe) HIP call; nothing else is shared.
struct DevOnce {
    std::atomic<uint64_t> bits{0};
};
static inline bool reserve_lds_once(DevOnce& once, const void* fn, int bytes) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    const uint64_t bit = 1ull << (dev & 63);
    if (once.bits.load(std::memory_order_acquire) & bit) return true;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return false;
    once.bits.fetch_or(bit, std::memory_order_release);
    return true;
}

#define STAR_REQUIRE(cond, ...)                         \
    do {                                                \
        if (!(cond)) {                                  \
            ::star::set_error(__VA_ARGS__);             \
            return STAR_ERR_ARG;                        \
        }                                               \
    } while (0)

#define STAR_LAUNCH_CHECK(what)                                                      \
    do {                                                                             \
        hipError_t e_ = hipGetLastError();                                           \
        if (e_ != hipSuccess) {                                                      \
            ::star::set_error("%s: launch failed: %s", what, hipGetErrorString(e_)); \
            return STAR_ERR_LAUNCH;                                                  \
        }                                                                            \
    } while (0)

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

#define LDS_PTR(T, p) ((__attribute__((address_space(3))) T*)(p))

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// wave-wide sum over 64 lanes (all lanes get the result)
template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// round-to-nearest-even fp32 -> bf16 bits / back (hipcc lowers the casts to v_cvt_pk_bf16_f32)
__device__ __forceinline__ __bf16 f2bf(float f) { return (__bf16)f; }
__device__ __forceinline__ float bf2f(__bf16 b) { return (float)b; }

static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// The serial integrate-and-fire chain of one utterance, run by ONE lane on a row resident in LDS (16-byte aligned,
// length a multiple of 16, zero padded): I += a_t; fire = I >= thr_t; I -= 1 where fired.  Every instruction of the
// lane's wave costs a full issue slot whatever the number of active lanes, so the loop holds only what is serial:
// per frame add, subtract, compare, select; the row is OVERWRITTEN with the pre-fire integrals I_t (after the add),
// from which all threads afterwards recover fires (I_t >= thr_t), positions and weights in parallel.
// HAS_THR = false: the constant threshold thr_const for every frame (thr_row unused).
// cur is the integrator carried in (0 at the start of an utterance); the carry after the last frame is returned.
template <bool HAS_THR>
__device__ __forceinline__ float cif_serial_chain(float* row, const float* thr_row, float thr_const, int nblk16,
                                                  float cur = 0.0f) {
    f32x4 a[4], th[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        a[q] = *(const f32x4*)(row + 4 * q);
        if constexpr (HAS_THR) th[q] = *(const f32x4*)(thr_row + 4 * q);
        else th[q] = (f32x4){thr_const, thr_const, thr_const, thr_const};
    }
    for (int blk = 0; blk < nblk16; ++blk) {
        // the next block's operands are requested before this block's chain starts
        f32x4 an[4], tn[4];
        const int nb = (blk + 1 < nblk16) ? blk + 1 : blk;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            an[q] = *(const f32x4*)(row + 16 * nb + 4 * q);
            if constexpr (HAS_THR) tn[q] = *(const f32x4*)(thr_row + 16 * nb + 4 * q);
            else tn[q] = th[q];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                cur = __fadd_rn(cur, a[q][j]);
                a[q][j] = cur;
                const float less = __fsub_rn(cur, 1.0f);
                cur = (cur >= th[q][j]) ? less : cur;
            }
#pragma unroll
        for (int q = 0; q < 4; ++q) *(f32x4*)(row + 16 * blk + 4 * q) = a[q];
#pragma unroll
        for (int q = 0; q < 4; ++q) { a[q] = an[q]; th[q] = tn[q]; }
    }
    return cur;
}

// The same chain for the common case -- threshold 1 on every frame, every weight in {0} u [FLT_MIN, 1) (1.0 itself
// EXCLUDED), carry in {0} u [FLT_MIN, 1): then I = cur + a <= 2 - 2^-23 is formed without rounding up to 2, lies in
// [0, 2), and "I >= 1 ? I - 1 : I" is fract(I) EXACTLY (I - 1 is exact for I in [1, 2) by Sterbenz, no denormal can
// appear).  With a weight of exactly 1.0 and a carry of 1 - 2^-24 the sum would round (ties to even) to 2.0, where the
// reference keeps a carry of 1.0 and fires again on the next frame while fract(2.0) = 0 does not: rows that hold an
// exact 1.0 therefore take the generic chain above.  Two dependent instructions per frame (v_add_f32, v_fract_f32)
// instead of four.  The caller checks the precondition; a forced fire (threshold -inf) is legal only where every later
// weight is zero, where the two recurrences can no longer produce different fires.
__device__ __forceinline__ float cif_serial_chain_unit(float* row, int nblk16, float cur) {
    f32x4 a[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) a[q] = *(const f32x4*)(row + 4 * q);
    for (int blk = 0; blk < nblk16; ++blk) {
        f32x4 an[4];
        const int nb = (blk + 1 < nblk16) ? blk + 1 : blk;
#pragma unroll
        for (int q = 0; q < 4; ++q) an[q] = *(const f32x4*)(row + 16 * nb + 4 * q);
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float t = __fadd_rn(cur, a[q][j]);
                a[q][j] = t;
                cur = __builtin_amdgcn_fractf(t);
            }
#pragma unroll
        for (int q = 0; q < 4; ++q) *(f32x4*)(row + 16 * blk + 4 * q) = a[q];
#pragma unroll
        for (int q = 0; q < 4; ++q) a[q] = an[q];
    }
    return cur;
}

// The two-instruction chain split in two: the serial lane keeps nothing but the carry INTO every block of 16 frames
// (carry[blk]; per 16 frames 4 ds_read_b128 and one ds_write_b32 beside the 32 dependent VALU instructions: 12.3 cycles
// per frame, the dependent-issue floor of v_add_f32 -> v_fract_f32, against 19.1 for the loop above, whose four
// ds_write_b128 per block take 13 cycles each on the store path -- profiles/r03_fire_chain_microbench.txt), and the
// workgroup's threads afterwards rebuild the pre-fire integrals of all blocks at once, one block per thread, with the
// same two instructions on the same operands: the row ends up bit-identical to cif_serial_chain_unit's.
__device__ __forceinline__ float cif_serial_carries_unit(const float* row, float* carry, int nblk16, float cur) {
    f32x4 a[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) a[q] = *(const f32x4*)(row + 4 * q);
    for (int blk = 0; blk < nblk16; ++blk) {
        f32x4 an[4];
        const int nb = (blk + 1 < nblk16) ? blk + 1 : blk;
#pragma unroll
        for (int q = 0; q < 4; ++q) an[q] = *(const f32x4*)(row + 16 * nb + 4 * q);
        carry[blk] = cur;
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int j = 0; j < 4; ++j) cur = __builtin_amdgcn_fractf(__fadd_rn(cur, a[q][j]));
#pragma unroll
        for (int q = 0; q < 4; ++q) a[q] = an[q];
    }
    return cur;
}
// block `blk` of the row: weights -> pre-fire integrals, from the carry into the block
__device__ __forceinline__ void cif_expand_block_unit(float* row, const float* carry, int blk) {
    float cur = carry[blk];
    f32x4 a[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) a[q] = *(const f32x4*)(row + 16 * blk + 4 * q);
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float t = __fadd_rn(cur, a[q][j]);
            a[q][j] = t;
            cur = __builtin_amdgcn_fractf(t);
        }
#pragma unroll
    for (int q = 0; q < 4; ++q) *(f32x4*)(row + 16 * blk + 4 * q) = a[q];
}

// exclusive scan of one int per thread over a workgroup of NW waves (wtot: NW ints of LDS); `total` = workgroup sum
template <int NW>
__device__ __forceinline__ int block_excl_scan(int v, int* wtot, int& total) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    int incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int n = __shfl_up(incl, o, 64);
        if (lane >= o) incl += n;
    }
    if (lane == 63) wtot[wid] = incl;
    __syncthreads();
    int base = 0, all = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        const int t = wtot[w];
        if (w < wid) base += t;
        all += t;
    }
    __syncthreads();
    total = all;
    return base + incl - v;
}
__device__ __forceinline__ int block_excl_scan_256(int v, int* wtot, int& total) {
    return block_excl_scan<4>(v, wtot, total);
}

}  // namespace star
