#!/bin/bash
# kernel-by-kernel timeline of ONE replayed pass of the shared-K/V cross-attention stack with the deferred weight
# gradients (tools/f1_stack.py --shared-only): rocprofv3 kernel trace -> tools/step_timeline.py
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03
timeout -k 10 300 rocprofv3 --kernel-trace -d gpurun_out/prof_f1t -o f1t --output-format csv -- python3 tools/f1_stack.py --shared-only > gpurun_out/prof_f1t.log 2>&1
echo exit $?
python3 tools/step_timeline.py gpurun_out/prof_f1t/f1t_kernel_trace.csv "gemm_nt_kernel<3" 2
