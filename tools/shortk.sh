S="dgrad:4,128,228,64,256,1,1 fwd:4,128,228,256,64,1,1 dgrad:4,128,228,256,64,1,1 fwd:4,128,228,64,256,1,1 dgrad:4,64,114,128,512,1,1 fwd:4,64,114,512,128,1,1"
for t in auto 64,64,1 128,64,1 128,128,1; do
  if [ $t = auto ]; then unset DPFT_FORCE_TILE; else export DPFT_FORCE_TILE=$t; fi
  timeout 200 python tools/conv_bench.py $S 2>&1 | grep -v amdgpu | awk -v t=$t '{printf "%s ", $3} END {print " <- " t}'
done
