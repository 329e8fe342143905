for f in auto 128,2 128,4 128,7 64,2 64,4 64,7 64,14 64,28; do
if [ $f = auto ]; then unset DPFT_FORCE_WGRAD; else export DPFT_FORCE_WGRAD=$f; fi
timeout 120 python tools/wgrad_small_bench.py 2>&1 | grep -v amdgpu | awk '{printf "%s %s %s | ", $1, $2, $3} END {print ""}'
done
