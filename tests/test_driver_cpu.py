"""CPU suite part 3: host driver logic — sharding, the hit vector and its all-reduce over 2 gloo ranks
(the N>1 path of bench.py with the oracle standing in for the kernels), formatters, CLI argument errors."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, load_golden_reads
from fulgor_amd import driver, pack_reads


def test_shard_ranges_are_contiguous_and_cover():
    for n in (0, 1, 7, 8, 1000003):
        for world in (1, 2, 3, 8):
            rs = [driver.shard_range(n, r, world) for r in range(world)]
            assert rs[0][0] == 0 and rs[-1][1] == n
            assert all(rs[i][1] == rs[i + 1][0] for i in range(world - 1))
            assert all(a <= b for a, b in rs)


def test_ascii_formatter_matches_reference_formatter(s10_oracle):
    reads = load_golden_reads()[:200]
    b, o = pack_reads(reads)
    offs, cols = s10_oracle.full_intersection(b, o)
    assert driver.format_ascii(17, offs, cols) == s10_oracle.format_ascii(offs, cols, first_id=17)
    # and the golden file itself is in that format
    want = open(os.path.join(GOLDEN, "s10_full_intersection.tsv"), "rb").read().splitlines(keepends=True)[:200]
    assert driver.format_ascii(0, offs, cols) == b"".join(want)


def test_binary_formatter_layout():
    offs = np.array([0, 2, 2], dtype=np.uint64)
    cols = np.array([5, 9], dtype=np.uint32)
    raw = driver.format_binary(3, offs, cols)
    assert np.frombuffer(raw, dtype="<u4").tolist() == [3, 2, 5, 9, 4, 0]


WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch, torch.distributed as dist
from fulgor_amd import driver, pack_reads
from fulgor_amd.reads import parse_fastx
from oracle.pyoracle import OracleIndex
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
reads = parse_fastx(os.path.join(sys.argv[1], "tests", "golden", "s10_reads.fa"))
lo, hi = driver.shard_range(len(reads), rank, world)
orc = OracleIndex.from_dump(os.path.join(sys.argv[1], "data", "s10"))
b, o = pack_reads(reads[lo:hi])
offs, cols = orc.full_intersection(b, o, threads=2)
t = torch.from_numpy(driver.hit_vector(offs, cols, 10))
driver.all_reduce_hits(t)
if rank == 0:
    np.save(sys.argv[2], t.numpy())
dist.barrier()
dist.destroy_process_group()
'''


def test_two_rank_gloo_hit_reduction(s10_oracle, s10_dump, tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    out = tmp_path / "hits.npy"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                    "--master-addr", "127.0.0.1", "--master-port", "29541", str(script), ROOT, str(out)],
                   check=True, env=env, timeout=600)
    reads = load_golden_reads()
    b, o = pack_reads(reads)
    offs, cols = s10_oracle.full_intersection(b, o)
    want = driver.hit_vector(offs, cols, 10)
    assert np.array_equal(np.load(out), want)
    assert want[10] == len(reads)


def test_cli_argument_errors():
    from fulgor_amd import cli
    assert cli.main(["pseudoalign", "-i", "x", "-q", "y", "-o", "z", "-r", "1.5"]) == 1
    assert cli.main(["pseudoalign", "-i", "x", "-q", "y", "-o", "z", "-r", "0.5", "--deduplicate"]) == 1
    assert cli.main(["pseudoalign", "-i", "x", "-q", "y", "-o", "z", "--format", "weird"]) == 1
    assert cli.main(["pseudoalign", "-i", "x"]) == 1
    assert cli.main(["build"]) == 1
