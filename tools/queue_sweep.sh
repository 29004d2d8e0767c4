#!/bin/bash
# diagnostic: headline throughput WITHOUT straggler deferral against the number of HIP hardware queues and in-flight streams
for q in ${QUEUES:-12 16 20 24}; do for s in ${STREAMS:-64 96 128}; do
GPU_MAX_HW_QUEUES=$q python bench.py --steps ${STEPS:-400} --warmup 20 --streams $s --defer 0 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('queues $q streams $s: value %.3e ms/step %.3f kernel_avg %.2f ms agg_frac %.3f max_it %d' % (d['value'], d['ms_per_step'], d['roofline']['kernel_avg_ms'], d['roofline']['aggregate_frac_per_gpu'], d['config']['max_admm_iters_rank0']))"
done; done
