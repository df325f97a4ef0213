"""topay_amd: MI355X-native batched replacement of TopAY's (s, theta) trajectory NLP.

Importing the package asks the HIP runtime for 24 hardware queues per priority level unless the environment already says
otherwise: the solver launches its six N-buckets on separate streams, and HIP streams that land on the same hardware
queue (default pool: 4, shared with every other stream of the process) run one after the other.  The runtime reads the
variable once, at its first call, so this has to happen before anything initialises HIP in the process.  24 = three
contexts in flight (18 streams) plus room for the streams of torch and RCCL; 32 and more are time-sliced by the scheduler
firmware and throughput collapses.
"""
import os

os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
