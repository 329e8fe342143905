for f in auto 64,64,1 64,64,2 64,64,4 64,64,8 64,64,16 64,64,32 128,64,4 128,64,8; do
if [ $f = auto ]; then unset DPFT_FORCE_TILE; else export DPFT_FORCE_TILE=$f; fi
timeout 120 python tools/igemm_small_bench.py 2>&1 | grep -v amdgpu | awk '{printf "%s %s %s | ", $1, $2, $3} END {print ""}'
done
