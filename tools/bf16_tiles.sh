S="fwd:8,32,57,256,256,3,1 dgrad:8,32,57,256,256,3,1 fwd:8,32,57,256,1024,1,1 fwd:8,32,57,1024,256,1,1 dgrad:8,32,57,1024,256,1,1 fwd:8,64,114,128,128,3,1 fwd:8,128,228,64,64,3,1 fwd:8,64,114,128,512,1,1"
for t in auto 64,64,1 128,64,1 128,128,1; do
  if [ $t = auto ]; then unset DPFT_FORCE_TILE; else export DPFT_FORCE_TILE=$t; fi
  DPFT_COMPUTE=bf16 timeout 200 python tools/conv_bench.py $S 2>&1 | grep -v amdgpu | awk -v t=$t '{printf "%s ", $3} END {print " <- " t}'
done
