// Dev tool: what does one frame of the integrate-and-fire recurrence cost a single wave?
//   chain A  t = cur + a; cur = fract(t)                       (cif_serial_chain_unit, operands in registers)
//   chain B  t = cur + a; cur = t >= 1 ? t - 1 : t             (the generic chain's add / sub / cmp / select)
//   chain C  dependent v_add_f32 only        chain D  dependent v_fract_f32 only
//   chain E  the library's loop itself: cif_serial_chain_unit on a row in LDS (loads, stores and waits included)
// Each chain runs N frames in one wave of one workgroup (the segmenter's situation: the other 15 waves of its workgroup
// wait at a barrier) and is timed with s_memtime; cycles per frame = delta / N.  Build and run:
//   hipcc --offload-arch=gfx950 -O3 -I star_amd/csrc -o tools/chain_microbench tools/chain_microbench.hip
//   tools/chain_microbench            (on the GPU box)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "common.h"

constexpr int N = 3072;          // frames (configs[3]: T = 3000)

template <int KIND>
__global__ __launch_bounds__(64) void chain_kernel(const float* __restrict__ a, float* __restrict__ out,
                                                   unsigned long long* __restrict__ cycles) {
    __shared__ __attribute__((aligned(16))) float row[N];
    for (int i = threadIdx.x; i < N; i += 64) row[i] = a[i];
    __syncthreads();
    float cur = KIND == 3 ? a[0] + 0.3f : 0.f;
    // every lane runs the chain (an instruction costs the wave the same whatever the number of active lanes); the timers
    // are tied to the chain's carry through asm operands so that nothing moves across them
    unsigned long long t0, t1, r0, r1;
    asm volatile("s_memrealtime %0\n\ts_memtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(r0), "=s"(t0) : "v"(cur));
    asm volatile("" : "+v"(cur) : "s"(t0));
    if (KIND == 4) {
        cur = star::cif_serial_chain_unit(row, N / 16, cur);
    } else {
        // operands in registers, 16 frames per block as in the library (a few hot LDS rows, read a block ahead)
        for (int blk = 0; blk < N / 16; ++blk) {
            float v[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) v[j] = row[(blk & 7) * 16 + j];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                if (KIND == 0) { const float t = __fadd_rn(cur, v[j]); cur = __builtin_amdgcn_fractf(t); }
                if (KIND == 1) { const float t = __fadd_rn(cur, v[j]); const float l = __fsub_rn(t, 1.0f); cur = t >= 1.0f ? l : t; }
                if (KIND == 2) { cur = __fadd_rn(cur, v[j]); }
                if (KIND == 3) { cur = __builtin_amdgcn_fractf(cur); asm volatile("" : "+v"(cur)); }
            }
        }
    }
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1) : "v"(cur));
    if (threadIdx.x == 0) {
        out[0] = cur;
        cycles[0] = t1 - t0;        // shader cycles (s_memtime)
        cycles[1] = r1 - r0;        // 100 MHz reference ticks (s_memrealtime)
    }
}

int main() {
    std::vector<float> h(N);
    unsigned s = 12345;
    for (int i = 0; i < N; ++i) {
        s = s * 1664525u + 1013904223u;
        h[i] = (float)(s >> 8) / 16777216.0f * 0.25f;          // weights in [0, 0.25): rate-4 compression
    }
    float *a, *out;
    unsigned long long* cyc;
    hipMalloc(&a, N * 4);
    hipMalloc(&out, 4);
    hipMalloc(&cyc, 16);
    hipMemcpy(a, h.data(), N * 4, hipMemcpyHostToDevice);
    const char* names[5] = {"A add+fract (registers)", "B add,sub,cmp,select (registers)", "C dependent v_add_f32",
                            "D dependent v_fract_f32", "E cif_serial_chain_unit on an LDS row"};
    for (int k = 0; k < 5; ++k) {
        unsigned long long best = ~0ull, best_ref = 0;
        float res = 0;
        for (int rep = 0; rep < 5; ++rep) {
            switch (k) {
                case 0: hipLaunchKernelGGL(chain_kernel<0>, dim3(1), dim3(64), 0, 0, a, out, cyc); break;
                case 1: hipLaunchKernelGGL(chain_kernel<1>, dim3(1), dim3(64), 0, 0, a, out, cyc); break;
                case 2: hipLaunchKernelGGL(chain_kernel<2>, dim3(1), dim3(64), 0, 0, a, out, cyc); break;
                case 3: hipLaunchKernelGGL(chain_kernel<3>, dim3(1), dim3(64), 0, 0, a, out, cyc); break;
                default: hipLaunchKernelGGL(chain_kernel<4>, dim3(1), dim3(64), 0, 0, a, out, cyc); break;
            }
            unsigned long long c[2];
            hipMemcpy(c, cyc, 16, hipMemcpyDeviceToHost);
            hipMemcpy(&res, out, 4, hipMemcpyDeviceToHost);
            if (c[0] < best) { best = c[0]; best_ref = c[1]; }
        }
        printf("%-40s %8llu cycles / %d frames = %5.1f cycles per frame, %5.2f ns per frame   [carry %.6f]\n", names[k], best,
               N, (double)best / N, best_ref * 10.0 / N, res);
    }
    return 0;
}
