S="${AB_SHAPES:-fwd:4,32,57,256,256,3,1 dgrad:4,32,57,256,256,3,1 wgrad:4,32,57,256,256,3,1 wgrad:4,32,57,256,1024,1,1 wgrad:4,32,57,1024,256,1,1 wgrad:4,128,228,64,64,3,1 wgrad:4,64,114,128,128,3,1 wgrad:4,16,29,512,512,3,1 wgrad:4,128,228,64,256,1,1 wgrad:4,64,114,512,128,1,1}"
cp dpft_amd/libdpft_hip.so /tmp/lib_keep.so
for v in old new old new; do cp tools/ab/lib_$v.so dpft_amd/libdpft_hip.so; echo "== $v"; timeout 300 python tools/conv_bench.py $S 2>&1 | grep -v amdgpu | awk '{printf "%s %s | ", $1, $3} END {print ""}'; done
cp /tmp/lib_keep.so dpft_amd/libdpft_hip.so
