cd /root/repo
for cfg in "0 0 0" "3 0 0" "4 0 0" "2 0 0" "0 3 0" "0 4 0" "0 2 0" "0 0 3" "0 0 4" "0 0 2" "0 0 8"; do
  set -- $cfg
  E=""
  [ "$1" != 0 ] && E="$E DPFT_SA_QW_FWD=$1"
  [ "$2" != 0 ] && E="$E DPFT_SA_QW_BWD=$2"
  [ "$3" != 0 ] && E="$E DPFT_SA_KW=$3"
  cd /tmp && export TMPDIR=/tmp
  env $E STEPS=3 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sa_$1$2$3 -- python /root/repo/tools/train_only.py </dev/null > /dev/null 2>&1
  f=$(find /tmp/sa_$1$2$3 -name "*kernel_stats.csv" | head -1)
  echo "== fwd=$1 bwd_q=$2 kv=$3: $(grep -E 'sa_train' $f | awk -F, '{gsub(/"/,"",$1); split($1,a,"<"); printf "%s<%s avg %.1f us | ", substr(a[1],12), substr(a[2],1,1), $4/1000}')"
done
