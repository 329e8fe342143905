# round 6: (a) the 128 x 64 epilogue-prefetch data-gradient kernel at 3 workgroups per CU (lib_v3: __launch_bounds__(256, 3), 168 VGPRs + 16 spilled)
# vs 2 (lib_v0: 190 VGPRs), isolated shape + whole step; (b) tile choice of the layer-3 1x1 forward / data-gradient shapes
cd /root/repo; export PYTHONUNBUFFERED=1
cp dpft_amd/libdpft_hip.so /tmp/lib_keep.so
step() { python bench.py --steps 40 --warmup 10 --no-cpu-baseline --latency-reps 5 2>/dev/null | grep "^{" | tail -n 1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['value'],1), 'mean', round(d['ms_per_step'],2), 'median', round(d['step_ms_median'],2), 'min', round(d['step_ms_min'],2), 'conv_ms', round(d['roofline']['conv_ms_per_step'],2))"; }
for rep in 1 2; do for v in 0 3; do
  cp tools/ab/lib_v$v.so dpft_amd/libdpft_hip.so
  DPFT_CONV_TABLE=gpurun_out/r6occ_table_v${v}_$rep.txt step "lib_v$v rep$rep"
done; done
cp /tmp/lib_keep.so dpft_amd/libdpft_hip.so
for v in 0 3; do echo "== table rows lib_v$v"; grep -h "^dgrad 4 32 57 1024 256 1 1\|^dgrad 4 64 114 512 128 1 1\|^dgrad 4 128 228 256 64 1 1\|^dgrad 4 16 29 2048 512 1 1" gpurun_out/r6occ_table_v${v}_*.txt; done
echo "== tile choice, isolated (auto vs forced)"
for spec in fwd:4,32,57,256,1024,1,1 fwd:4,32,57,1024,256,1,1 dgrad:4,32,57,256,1024,1,1 dgrad:4,32,57,1024,256,1,1; do
  python tools/conv_bench.py $spec
  for t in 128,64,1 128,128,1 64,64,1; do DPFT_FORCE_TILE=$t python tools/conv_bench.py $spec; done
done
