#!/usr/bin/env python
"""Started by tools/launch_node.sh in place of the script: binds this rank to its contiguous share of the host cores (LOCAL_RANK x
SP_CORES_PER_RANK ...), then runs the script in this process.  No-op where sched_setaffinity is missing or the share is empty."""
import os
import runpy
import sys

rank, per = int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("SP_CORES_PER_RANK", "0"))
if per > 0 and hasattr(os, "sched_setaffinity"):
    allowed = sorted(os.sched_getaffinity(0))
    mine = allowed[rank * per: (rank + 1) * per]
    if mine:
        os.sched_setaffinity(0, mine)
script, sys.argv = sys.argv[1], sys.argv[1:]
sys.path.insert(0, os.path.dirname(os.path.abspath(script)))
runpy.run_path(script, run_name="__main__")
