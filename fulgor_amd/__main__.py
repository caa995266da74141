import os
import sys

from .cli import main

rc = main()
# One command, one process: everything is written and closed when main() returns. Returning through the interpreter's and the HIP
# runtime's teardown — unpinning half a gigabyte of host buffers, freeing gigabytes of device memory buffer by buffer — took 0.3 s
# of a 1.0 s command; the operating system reclaims all of it at once. (FULGOR_ORDERLY_EXIT=1: the long way, for leak checkers.)
sys.stdout.flush()
sys.stderr.flush()
if os.environ.get("FULGOR_ORDERLY_EXIT"):
    sys.exit(rc)
os._exit(rc or 0)
