S="fwd:4,32,57,256,256,3,1 dgrad:4,32,57,256,256,3,1 fwd:4,32,57,1024,256,1,1 dgrad:4,32,57,1024,256,1,1 fwd:4,32,57,256,1024,1,1 dgrad:4,32,57,256,1024,1,1 fwd:4,64,114,128,128,3,1 fwd:4,128,228,64,256,1,1 dgrad:4,128,228,64,256,1,1 fwd:4,64,114,128,512,1,1"
python tools/conv_bench.py $S 2>&1 | grep -v amdgpu
for t in 64,64,1 128,64,1 128,128,1 128,128,2 128,64,2 128,128,3; do DPFT_FORCE_TILE=$t python tools/conv_bench.py $S 2>&1 | grep -v amdgpu; done
