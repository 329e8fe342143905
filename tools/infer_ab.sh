#!/bin/bash
# inference: K-split convs finish with the BatchNorm / residual / ReLU epilogue inside the launch (DPFT_BNACT_FIXUP=1, default) vs
# conv + reduction + elementwise pass (0)   -> gpurun_out/infer_ab.txt
{
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_gpu_conv_table.py -x -q 2>&1 | tail -3
for r in 1 2 3; do for v in 1 0; do for b in 1 4; do
  echo "fixup=$v batch=$b $(DPFT_BNACT_FIXUP=$v BATCH=$b REPS=200 python tools/infer_only.py 2>/dev/null | tail -1)"
done; done; done
} > gpurun_out/infer_ab.txt 2>&1
