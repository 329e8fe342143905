# usage: bash tools/ab_cmd.sh '<command>'  -- runs it alternately against tools/ab/lib_old.so and lib_new.so
cp dpft_amd/libdpft_hip.so /tmp/lib_keep.so
for v in old new old new; do cp tools/ab/lib_$v.so dpft_amd/libdpft_hip.so; echo "== $v"; eval "$1" 2>&1 | grep -v amdgpu; done
cp /tmp/lib_keep.so dpft_amd/libdpft_hip.so
