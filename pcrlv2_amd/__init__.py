"""pcrlv2_amd: the PCRLv2 pre-training step on MI355X (gfx950).  See DESIGN.md."""
import os as _os

# A training step drives three HIP streams (main, the second view, the weight-gradient / side-branch stream -- which also carries the
# data-parallel wrapper's bucket sums and collectives); RCCL adds its own.  That is exactly ROCm's default of 4 hardware queues per process
# (GPU_MAX_HW_QUEUES), which is deliberately NOT raised: measured on one MI355X with a one-rank RCCL process group (tools/pg_overhead_probe.sh),
# 8 queues are neutral without RCCL (32.9 vs 33.3 ms per step) and cost 7.5 ms per step as soon as a process group exists (40.7 vs 33.1 ms:
# round 2 set 8 by default and never ran the multi-rank path on a GPU).  An explicit GPU_MAX_HW_QUEUES in the environment is respected.
