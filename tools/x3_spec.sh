cd /root/repo
export PYTHONUNBUFFERED=1
S="fwd:4,32,57,256,256,3,1 dgrad:4,32,57,256,256,3,1 fwd:4,64,114,128,128,3,1 fwd:4,16,29,512,512,3,1 fwd:4,128,228,64,64,3,1 fwd:4,32,57,1024,256,1,1 fwd:4,32,57,256,1024,1,1 dgrad:4,32,57,256,1024,1,1"
for t in 128,64,1 128,64,2 64,64,1 64,64,2; do
for sp in 0 1; do
echo "== tile $t spec $sp"; DPFT_X3_SPEC=$sp DPFT_COMPUTE=bf16x3 DPFT_FORCE_TILE=$t python tools/conv_bench.py $S 2>&1 | grep -v amdgpu
done; done
DPFT_X3_SPEC=1 DPFT_CONV_COMPUTE=bf16x3 DPFT_FORCE_TILE=128,64,1 python -m pytest tests/test_gpu_conv_table.py -q -x -k "conv_problem_vs_fp64 and (32x57x256x256 or 64x114x128x128 or 16x29x512x512)" 2>&1 | tail -n 3
