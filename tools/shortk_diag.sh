cd /root/repo
export PYTHONUNBUFFERED=1
S="fwd:4,128,228,64,256,1,1 fwd:4,64,114,128,512,1,1 fwd:4,128,228,256,64,1,1 dgrad:4,128,228,64,256,1,1 dgrad:4,128,228,256,64,1,1 dgrad:4,64,114,512,128,1,1"
echo "== default (prologue + statistics)"; python tools/conv_bench.py $S 2>&1 | grep -v amdgpu
echo "== no statistics"; NOSTATS=1 python tools/conv_bench.py $S 2>&1 | grep -v amdgpu
echo "== no prologue"; NOPRO=1 python tools/conv_bench.py $S 2>&1 | grep -v amdgpu
echo "== neither"; NOPRO=1 NOSTATS=1 python tools/conv_bench.py $S 2>&1 | grep -v amdgpu
echo "== neither, 64x64"; DPFT_FORCE_TILE=64,64,1 NOPRO=1 NOSTATS=1 python tools/conv_bench.py $S 2>&1 | grep -v amdgpu
echo "== neither, 128x128"; DPFT_FORCE_TILE=128,128,1 NOPRO=1 NOSTATS=1 python tools/conv_bench.py $S 2>&1 | grep -v amdgpu
