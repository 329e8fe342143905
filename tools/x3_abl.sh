cd /root/repo
export PYTHONUNBUFFERED=1 NOCHECK=1 DPFT_FORCE_TILE=128,128,1
for abl in 0 1 2 3 4 8 11 12; do
  echo "ABL=$abl"
  DPFT_X3_ABL=$abl python tools/x3_planes_check.py fwd:4,64,114,128,128,3,1 fwd:4,32,57,256,256,3,1 fwd:8,64,114,128,128,3,1 2>&1 | grep -v amdgpu.ids
done > gpurun_out/x3_abl.txt
