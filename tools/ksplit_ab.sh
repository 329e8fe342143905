S="fwd:4,32,57,256,256,3,1 dgrad:4,32,57,256,256,3,1 fwd:4,32,57,256,1024,1,1 fwd:4,32,57,1024,256,1,1 dgrad:4,32,57,1024,256,1,1 dgrad:4,32,57,256,1024,1,1 fwd:4,16,29,512,512,3,1 dgrad:4,16,29,512,512,3,1 fwd:4,16,29,2048,512,1,1 fwd:4,64,114,128,128,3,1 fwd:4,64,114,512,128,1,1"
for k in 0 1 0 1; do
  DPFT_KSPLIT=$k timeout 200 python tools/conv_bench.py $S 2>&1 | grep -v amdgpu | awk -v t=$k '{printf "%s ", $3} END {print " <- ksplit " t}'
done
