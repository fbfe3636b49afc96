# strong-scaling line (one whole 993-query evaluation per step) under a few settings, alternated: tools/strong_ab.sh
for rep in 1 2; do
for cfg in "GPU_MAX_HW_QUEUES=8" "GPU_MAX_HW_QUEUES=4" "GPU_MAX_HW_QUEUES=8 CORSAIR_REGISTER_FRESH_THREADS=1" "GPU_MAX_HW_QUEUES=4 CORSAIR_REGISTER_FRESH_THREADS=1"; do
env $cfg python bench.py --scaling strong --steps 4 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('[$cfg]', round(d['value'],1), 'q/s', round(d['ms_per_step'],1), 'ms per evaluation')"
done
done
