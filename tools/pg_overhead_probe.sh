#!/bin/bash
# What does an initialised RCCL process group / the data-parallel wrapper cost a ONE-GPU step, and which runtime setting decides it?
# (bench.py PCRL_FORCE_DDP: 2 = the process group alone, 1 = the wrapper on a one-rank RCCL group: everything but the wire.)
#   gpurun -- tools/pg_overhead_probe.sh
run() { env "$@" python bench.py --steps 20 --warmup 6 --no-cpu-baseline --no-alone --no-secondary 2>/dev/null | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('%-78s -> %.2f ms per step' % ('$*', d['ms_per_step']))"; }
for q in 4 8 2; do
  run GPU_MAX_HW_QUEUES=$q
  run GPU_MAX_HW_QUEUES=$q PCRL_FORCE_DDP=2
  run GPU_MAX_HW_QUEUES=$q PCRL_FORCE_DDP=1 PCRL_DDP_OVERLAP=0
  run GPU_MAX_HW_QUEUES=$q PCRL_FORCE_DDP=1 PCRL_DDP_OVERLAP=1
done
