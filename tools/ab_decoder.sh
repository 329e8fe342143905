# same-box A/B of library builds under tools/ab/: bash tools/ab_decoder.sh a.so b.so ...
cp /root/repo/dpft_amd/libdpft_hip.so /tmp/lib_keep.so
for round in 1 2 3; do
  for lib in "$@"; do
    cp /root/repo/tools/ab/$lib /root/repo/dpft_amd/libdpft_hip.so
    echo "$lib $(REPS=200 timeout 60 python /root/repo/tools/decoder_only.py | tail -1)"
  done
done
cp /tmp/lib_keep.so /root/repo/dpft_amd/libdpft_hip.so
