// NuggetStep HARNESS kernels -- not part of the product path and not reference operators (include/star_harness.h).
// The step that bench.py / smoke() / the whole-step tests run derives the per-layer K_l / V_l
// from the gathered nugget memory with a fixed diagonal linear map standing in for the decoder's c_attn GEMM (row a9,
// which stays on hipBLASLt through torch); these two kernels keep that stand-in at one pass over the data.
#include "common.h"
#include "../../include/star_harness.h"

namespace star {

// ---- diagonal K/V fan-out (stand-in for the per-layer c_attn projection in the NuggetStep harness) ----
// out[j][r][:] = mem[r][:] * scales[j][:]   and   dmem[r][:] = sum_j g_j[r][:] * scales[j][:]
constexpr int FAN_MAX = 16;
struct FanPtrs {
    const __bf16* p[FAN_MAX];
};

__global__ __launch_bounds__(256) void diag_fanout_fwd_kernel(const __bf16* __restrict__ mem,
                                                              const __bf16* __restrict__ scales, int J, int64_t nvec,
                                                              int vec_per_row, __bf16* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= nvec) return;
    const int c8 = (int)(i % vec_per_row);
    const bf16x8 m = ((const bf16x8*)mem)[i];
    for (int j = 0; j < J; ++j) {
        const bf16x8 sc = ((const bf16x8*)scales)[(int64_t)j * vec_per_row + c8];
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (__bf16)((float)m[e] * (float)sc[e]);
        ((bf16x8*)out)[(int64_t)j * nvec + i] = o;
    }
}

__global__ __launch_bounds__(256) void diag_fanout_bwd_kernel(FanPtrs g, const __bf16* __restrict__ scales, int J,
                                                              int64_t nvec, int vec_per_row,
                                                              __bf16* __restrict__ dmem) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= nvec) return;
    const int c8 = (int)(i % vec_per_row);
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int j = 0; j < J; ++j) {
        const bf16x8 x = ((const bf16x8*)g.p[j])[i];
        const bf16x8 sc = ((const bf16x8*)scales)[(int64_t)j * vec_per_row + c8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += (float)x[e] * (float)sc[e];
    }
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (__bf16)acc[e];
    ((bf16x8*)dmem)[i] = o;
}

}  // namespace star

extern "C" int star_diag_fanout_fwd(const void* mem, const void* scales, int J, int64_t R, int d, void* out,
                                    void* stream) {
    using namespace star;
    STAR_REQUIRE(mem && scales && out, "star_diag_fanout_fwd: null pointer");
    STAR_REQUIRE(J >= 1 && J <= FAN_MAX && R >= 1 && d >= 8 && d % 8 == 0, "star_diag_fanout_fwd: bad shape J=%d R=%lld d=%d",
                 J, (long long)R, d);
    const int64_t nvec = R * (d / 8);
    hipLaunchKernelGGL(diag_fanout_fwd_kernel, dim3(cdiv(nvec, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const __bf16*)mem, (const __bf16*)scales, J, nvec, d / 8, (__bf16*)out);
    STAR_LAUNCH_CHECK("star_diag_fanout_fwd");
    return STAR_OK;
}

extern "C" int star_diag_fanout_bwd(const void* const* grads_host, int J, const void* scales, int64_t R, int d,
                                    void* dmem, void* stream) {
    using namespace star;
    STAR_REQUIRE(grads_host && scales && dmem, "star_diag_fanout_bwd: null pointer");
    STAR_REQUIRE(J >= 1 && J <= FAN_MAX && R >= 1 && d >= 8 && d % 8 == 0, "star_diag_fanout_bwd: bad shape J=%d R=%lld d=%d",
                 J, (long long)R, d);
    FanPtrs g = {};
    for (int j = 0; j < J; ++j) {
        STAR_REQUIRE(grads_host[j], "star_diag_fanout_bwd: grads[%d] is null", j);
        g.p[j] = (const __bf16*)grads_host[j];
    }
    const int64_t nvec = R * (d / 8);
    hipLaunchKernelGGL(diag_fanout_bwd_kernel, dim3(cdiv(nvec, 256)), dim3(256), 0, (hipStream_t)stream, g,
                       (const __bf16*)scales, J, nvec, d / 8, (__bf16*)dmem);
    STAR_LAUNCH_CHECK("star_diag_fanout_bwd");
    return STAR_OK;
}

