S="fwd:4,128,228,64,64,3,1 dgrad:4,128,228,64,64,3,1 fwd:4,128,228,64,256,1,1"
cp dpft_amd/libdpft_hip.so /tmp/lib_keep.so
for t in 64,64,1 128,64,1; do for v in old new; do cp tools/ab/lib_$v.so dpft_amd/libdpft_hip.so; echo "== $v tile $t: $(DPFT_FORCE_TILE=$t timeout 300 python tools/conv_bench.py $S 2>&1 | grep -v amdgpu | awk '{printf "%s %s | ", $1, $3}')"; done; done
cp /tmp/lib_keep.so dpft_amd/libdpft_hip.so
