# same-box A/B of the working tree against the python tree exported to ab_old/ (git archive HEAD ... | tar -x -C ab_old; the
# built library is shared): alternating bench runs, samples/s and median ms per step
cd /root/repo
for r in 1 2 3; do for w in old new; do
  d=/root/repo; [ $w = old ] && d=/root/repo/ab_old
  (cd $d && timeout 300 python bench.py --steps ${STEPS:-100} --warmup 20 --no-cpu-baseline --latency-reps 5 $BENCH_ARGS 2>/dev/null | python -c "
import json,sys; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w', round(b['value'],2), 'samples/s', round(b['ms_per_step'],3), 'ms/step', 'median', b.get('ms_per_step_median'))")
done; done
