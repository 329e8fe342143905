S="wgrad:4,32,57,256,256,3,1 wgrad:4,32,57,256,1024,1,1 wgrad:4,32,57,1024,256,1,1 wgrad:4,64,114,128,128,3,1 wgrad:4,128,228,64,64,3,1 wgrad:4,64,114,128,512,1,1 wgrad:4,128,228,64,256,1,1 wgrad:4,16,29,512,512,3,1"
python tools/conv_bench.py $S 2>&1 | grep -v amdgpu
for t in 128,6 128,7 128,8 128,14 128,16 128,28 128,32 128,56 64,8 64,12 64,16 64,28 64,32; do echo "== $t"; DPFT_FORCE_WGRAD=$t python tools/conv_bench.py $S 2>&1 | grep -v amdgpu | cut -c1-34,52-80; done
