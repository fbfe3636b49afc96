# A/B of the round-4 back-half fusions on one box: tools/ab_fuse.sh [workload]  (sequential value, then three batches in flight)
wl=${1:-chair}
for rep in 1 2 3; do
for v in 0 1; do
  env CS_RANSAC_FUSE_SCAN=$v CS_RANSAC_FUSE_S2LIST=$v python bench.py --workload $wl --no-cpu-baseline --no-overlap-probe --no-solo-probe --no-extra-workloads 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); c=d['config']; print('$wl fuse=$v sequential', round(d['value'],1), repr(c['ransac_mean_iters']), repr(c['rre_mean_deg']), (d.get('ransac_prefilter') or {}).get('survivors'), round(d['kernel_ms']['ransac_eval'],1))"
  env CS_RANSAC_FUSE_SCAN=$v CS_RANSAC_FUSE_S2LIST=$v python bench.py --workload $wl --pipeline 3 --steps 12 --warmup 6 --no-cpu-baseline --no-overlap-probe --no-solo-probe --no-extra-workloads 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$wl fuse=$v pipelined', round(d['value'],1))"
done
done
