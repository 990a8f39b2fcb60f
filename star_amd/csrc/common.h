// Shared helpers for the libstar_nugget.so kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/star_nugget.h"

namespace star {

void set_error(const char* fmt, ...);

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

}  // namespace star
