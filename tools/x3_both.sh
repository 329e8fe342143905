cd /root/repo
bash tools/x3_planes.sh
bash tools/x3_bench.sh quick
