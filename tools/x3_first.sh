set -x
cd /root/repo
export PYTHONUNBUFFERED=1
SPECS="fwd:4,32,57,256,256,3,1 dgrad:4,32,57,256,256,3,1 fwd:4,32,57,256,1024,1,1 fwd:4,32,57,1024,256,1,1 dgrad:4,32,57,1024,256,1,1 dgrad:4,32,57,256,1024,1,1 fwd:4,128,228,64,256,1,1 fwd:4,128,228,64,64,3,1 fwd:4,64,114,128,128,3,1 fwd:4,16,29,512,512,3,1 dgrad:4,128,228,256,64,1,1 dgrad:4,64,114,512,128,1,1"
python tools/conv_bench.py $SPECS > gpurun_out/x3_bench_fp32.txt 2>&1
DPFT_COMPUTE=bf16x3 python tools/conv_bench.py $SPECS > gpurun_out/x3_bench_x3.txt 2>&1
for t in 128,128,1 128,128,2 128,128,4 128,64,1 128,64,2 64,64,1; do
  DPFT_FORCE_TILE=$t DPFT_COMPUTE=bf16x3 python tools/conv_bench.py $SPECS >> gpurun_out/x3_bench_x3_tiles.txt 2>&1
done
DPFT_CONV_COMPUTE=bf16x3 timeout 900 python -m pytest tests/test_gpu_conv_table.py -q -s -k "conv_problem_vs_fp64" > gpurun_out/x3_table_x3.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_conv_table.py -q -s -k "conv_problem_vs_fp64" > gpurun_out/x3_table_fp32.txt 2>&1
tail -3 gpurun_out/x3_table_x3.txt gpurun_out/x3_table_fp32.txt
