"""One-off check of bench.py's eager fallback: graph capture made to fail, the line must still come out and say so."""
import io, json, sys, contextlib
sys.path.insert(0, ".")
sys.argv = ["bench.py", "--graph", "--steps", "4", "--warmup", "1", "--no-cpu-baseline", "--no-secondary", "--no-pmc"]
from surfacenetworks_amd import arap


class Boom:
    def __init__(self, *a, **k):
        raise RuntimeError("capture refused (test)")


arap.GraphedTrainStep = Boom
import bench
bench.main()          # (the line goes to the process's stdout: read it there)
