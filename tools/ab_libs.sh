#!/bin/bash
# usage: bash tools/ab_libs.sh "<v list>" '<command printing one line>'   -- alternates tools/ab/lib_v*.so builds, two rounds
cp dpft_amd/libdpft_hip.so /tmp/lib_keep.so
for rep in 1 2; do for v in $1; do cp tools/ab/lib_v$v.so dpft_amd/libdpft_hip.so; echo -n "v$v: "; eval "$2" 2>/dev/null | tail -1; done; done
cp /tmp/lib_keep.so dpft_amd/libdpft_hip.so
