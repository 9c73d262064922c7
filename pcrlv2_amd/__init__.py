"""pcrlv2_amd: the PCRLv2 pre-training step on MI355X (gfx950).  See DESIGN.md."""
import os as _os

# A training step uses up to five HIP streams (main, the second view, the weight-gradient / side-branch stream, the data-parallel wrapper's
# communication stream, RCCL's own).  ROCm maps streams onto 4 hardware queues by default; two streams on one queue serialize.  Ask for 8
# before the HIP runtime starts (no effect once it has; measured neutral on one GPU: 965.6 vs 968.0 crops/s same box).
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
