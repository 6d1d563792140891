"""What the kernels of this tree were built from: a hash of the kernel sources and the build flags. Counter files
(tools/collect_counters.sh -> profiles/rN/counters_<workload>.json) store it, and bench.py marks counters that were
collected from other kernels as stale instead of dividing them by today's durations."""
import hashlib
import os

HERE = os.path.dirname(os.path.abspath(__file__))
FILES = ("csrc/rt_kernels.hip", "csrc/rt_launch.h", "csrc/rt_device.h", "csrc/Makefile")


def kernels_hash():
    h = hashlib.sha256()
    for f in FILES:
        with open(os.path.join(HERE, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]
