cd /root/repo
export PYTHONUNBUFFERED=1
S="fwd:4,32,57,256,256,3,1 dgrad:4,32,57,256,256,3,1 fwd:4,64,114,128,128,3,1 fwd:4,16,29,512,512,3,1 fwd:4,128,228,64,64,3,1"
echo "== in order"; python tools/conv_bench.py $S 2>&1 | grep -v amdgpu
echo "== rotated"; DPFT_ABLATE=16 python tools/conv_bench.py $S 2>&1 | grep -v amdgpu
echo "== in order"; python tools/conv_bench.py $S 2>&1 | grep -v amdgpu
echo "== rotated"; DPFT_ABLATE=16 python tools/conv_bench.py $S 2>&1 | grep -v amdgpu
DPFT_ABLATE=16 python -m pytest tests/test_gpu_conv_x3.py -q -k "error_not_above" 2>&1 | tail -n 2
