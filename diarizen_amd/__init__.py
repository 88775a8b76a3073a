"""diarizen_amd — MI355X-native engine for the DiariZen sliding-window inference hot path."""
__version__ = "0.1.0"

import os as _os

# The engine overlaps independent batches on several HIP streams (inference.WindowRunner: one per engine handle) and the host stage
# has streams of its own (csrc/linkage.hip, postprocess.DevicePost).  The HIP runtime maps streams onto GPU_MAX_HW_QUEUES hardware
# queues (default 4) in creation order, and streams that share a queue serialise: measured on the bench process (r5), the
# three-handle configs[1] leg fell from 4.5 k to 3.7 k audio-s/s once earlier legs had created their streams, and came back with 8
# queues (profiles/r5_hw_queues_probe.txt).  Only a default: an explicit setting in the environment wins; it has to be in place
# before the first HIP call of the process, which importing this package before touching the device guarantees.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
