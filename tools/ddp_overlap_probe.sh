#!/bin/bash
# Where does launching buckets from inside backward (PCRL_DDP_OVERLAP=1) spend its time on ONE GPU (one-rank RCCL group: the collectives are free)?
#   gpurun -- tools/ddp_overlap_probe.sh
run() { env "$@" python bench.py --steps 20 --warmup 6 --no-cpu-baseline --no-alone --no-secondary 2>/dev/null | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('%-96s -> %.2f ms per step' % ('$*', d['ms_per_step']))"; }
for r in 1 2; do
run NO_WRAPPER=1
run PCRL_FORCE_DDP=1 PCRL_DDP_OVERLAP=0
run PCRL_FORCE_DDP=1 PCRL_DDP_OVERLAP=1
run PCRL_FORCE_DDP=1 PCRL_DDP_OVERLAP=1 PCRL_DDP_COMM_STREAM=own
run PCRL_FORCE_DDP=1 PCRL_DDP_OVERLAP=1 PCRL_DDP_PROBE_SKIP_COLLECTIVE=1
run PCRL_FORCE_DDP=1 PCRL_DDP_OVERLAP=1 PCRL_DDP_PROBE_DEFER=1
run PCRL_FORCE_DDP=1 PCRL_DDP_OVERLAP=1 PCRL_MAX_STEPS_AHEAD=4
done
