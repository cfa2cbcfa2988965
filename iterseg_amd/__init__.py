"""iterseg_amd -- iterseg's affinity-U-Net / watershed hot path on MI355X."""
import os as _os

# The flood runs its three size tiers side by side on three HIP streams (csrc/ws.hip).  The HIP runtime multiplexes a process's
# streams onto GPU_MAX_HW_QUEUES hardware queues (4 by default), and two streams that share a queue run one after the other: with a
# U-Net context created before the watershed context the tiers of a structured 512^3 volume took 96 ms instead of 58 ms (LOG.md
# round 5).  The variable is read when the runtime initialises, i.e. at the first HIP call of the process: import this package
# before touching the GPU, or export it yourself.  An explicit setting is respected.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
