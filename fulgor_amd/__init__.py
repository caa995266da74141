"""fulgor_amd — MI355X-native pseudoalignment query engine (drop-in for the `fulgor pseudoalign` hot
path of jermp/fulgor). Package contents: csrc/ (HIP kernels + C ABI + host index code), index.py (the
host-side mirror of the reference's index interface), reads.py (read sources)."""
from .index import (Index, Reads, Result, pack_reads, FULL_INTERSECTION, THRESHOLD_UNION,  # noqa: F401
                    HYBRID, DIFF, META, META_DIFF)
