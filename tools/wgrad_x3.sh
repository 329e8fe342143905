cd /root/repo
export PYTHONUNBUFFERED=1
SPECS="wgrad:4,32,57,256,256,3,1 wgrad:4,64,114,128,128,3,1 wgrad:4,16,29,512,512,3,1 wgrad:4,128,228,64,64,3,1 wgrad:4,64,114,256,256,3,2"
echo fp32; DPFT_WGRAD_X3=0 python tools/conv_bench.py $SPECS 2>&1 | grep -v amdgpu.ids
echo x3; DPFT_WGRAD_X3=1 python tools/conv_bench.py $SPECS 2>&1 | grep -v amdgpu.ids
timeout 900 python -m pytest tests/test_gpu_conv_table.py -q -x -k "conv_problem_vs_fp64" 2>&1 | tail -n 4
