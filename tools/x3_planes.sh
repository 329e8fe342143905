cd /root/repo
export PYTHONUNBUFFERED=1
python tools/x3_planes_check.py 2>&1 | grep -v amdgpu.ids > gpurun_out/x3p_auto.txt
rm -f gpurun_out/x3p_tiles.txt
for t in 128,128,1 128,128,2 128,128,4 128,64,1 128,64,2 64,64,1 64,64,2; do
  NOCHECK=1 DPFT_FORCE_TILE=$t python tools/x3_planes_check.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/x3p_tiles.txt
done
