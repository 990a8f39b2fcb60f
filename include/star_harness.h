/* star_harness.h -- HARNESS-ONLY entry points of libstar_nugget.so.
 *
 * Nothing here replaces a reference operator and no product caller needs it: these kernels exist for the NuggetStep
 * that bench.py, __graft_entry__.smoke() and the whole-step tests run, where the decoder's
 * c_attn projection (my_transformer.py:500-501, row a9: a GEMM that stays on hipBLASLt through torch) is replaced by
 * a fixed diagonal linear map so that the step contains the hot path and nothing else.  Conventions as in
 * star_nugget.h (device pointers, asynchronous on `stream`, 0 / negative STAR_ERR_* return). */
#ifndef STAR_HARNESS_H
#define STAR_HARNESS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Diagonal K/V fan-out: the per-layer K_l / V_l of the harness step from the gathered memory, one pass over the data.
 *   fwd: out[j][r][:] = mem[r][:] * scales[j][:]      mem [R,d], scales [J,d], out [J,R,d], all bf16, J <= 16
 *   bwd: dmem[r][:]   = sum_j grads[j][r][:] * scales[j][:]   (fp32 accumulate); grads_host is a HOST array
 *        of J device pointers, each a dense [R,d] bf16 buffer. */
int star_diag_fanout_fwd(const void* mem, const void* scales, int J, int64_t R, int d, void* out, void* stream);
int star_diag_fanout_bwd(const void* const* grads_host, int J, const void* scales, int64_t R, int d, void* dmem,
                         void* stream);

#ifdef __cplusplus
}
#endif
#endif /* STAR_HARNESS_H */
