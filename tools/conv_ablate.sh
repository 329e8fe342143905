#!/bin/bash
# Where does the time of the igemm kernels go?  DPFT_ABLATE bits: 1 no global loads, 2 no LDS stores (incl. the BN+ReLU prologue),
# 4 no epilogue, 8 no MFMAs / fragment reads.
S="fwd:4,32,57,256,256,3,1 dgrad:4,32,57,256,256,3,1 fwd:4,32,57,256,1024,1,1 fwd:4,32,57,1024,256,1,1 dgrad:4,32,57,1024,256,1,1 dgrad:4,32,57,256,1024,1,1 fwd:4,128,228,64,256,1,1 dgrad:4,128,228,256,64,1,1 fwd:4,64,114,128,128,3,1"
for ab in 0 4 1 3 7 6 8 12 15; do
  echo "== DPFT_ABLATE=$ab"
  DPFT_ABLATE=$ab python tools/conv_bench.py $S 2>&1 | grep -v amdgpu
done
