for q in 4 8 16; do
GPU_MAX_HW_QUEUES=$q python bench.py --no-cpu-baseline --no-solo-probe > gpurun_out/bq_$q.json 2> gpurun_out/bq_$q.err && python -c "
import json; d=json.load(open('gpurun_out/bq_$q.json')); print('GPU_MAX_HW_QUEUES=$q', round(d['value'],1), round(d['two_batches_in_flight']['value'],1), d['two_batches_in_flight']['identical_poses'])" || exit 1
done
