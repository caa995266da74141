"""CPU suite part 2: host logic of the engine (dump ingestion, hybrid encoder, dictionary, container),
the C ABI surface, and the loud failure of query entry points without a GPU."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

import fulgor_amd
from conftest import ROOT, has_gpu


@pytest.fixture(scope="module")
def host_index(s10_fgidx):
    return fulgor_amd.Index(s10_fgidx, device=-1)


def test_header_symbols_are_exported(built):
    hdr = open(os.path.join(ROOT, "include", "fulgor_gpu.h")).read()
    names = set(re.findall(r"\b(fgpu_[a-z_]+)\s*\(", hdr))
    assert len(names) >= 25
    lib = C.CDLL(built.LIB_GPU)
    missing = [n for n in sorted(names) if not hasattr(lib, n)]
    assert not missing, missing


def test_copy_engine_classification_rule(built):
    """round-5 review, item 3: which copy engines take the bulk copies is decided from a timing of every engine when a process
    starts — on an 8-GPU node by eight processes at once. The rule: the engines within half of the best rate; three or four of them
    is what the hardware has (50 against 7-13 GB/s); anything else is ambiguous (the library then uses its default engines)."""
    from fulgor_amd import _native
    L = _native.lib()

    def fast(rates):
        eng = (C.c_uint32 * len(rates))(*[1 << i for i in range(len(rates))])
        gb = (C.c_double * len(rates))(*rates)
        out, n = (C.c_uint32 * len(rates))(), C.c_uint32()
        assert L.fgpu_copy_engines_classify(eng, gb, len(rates), out, C.byref(n)) == 0
        return list(out[:n.value])

    quiet = [50, 50, 50, 50, 12, 11, 11, 11, 8, 10, 9, 9, 7, 7, 7, 7]          # as measured (profiles/r5/e2e_once_engines_r5.txt)
    assert fast(quiet) == [1, 2, 4, 8]
    assert fast([x * 0.6 for x in quiet[:2]] + quiet[2:]) == [1, 2, 4, 8]         # two engines slowed by a neighbour's traffic
    assert fast([50, 50, 50] + quiet[4:]) == [1, 2, 4]                            # three fast engines
    assert fast([20] * 16) == []                                                  # everything alike: ambiguous
    assert fast([50, 12, 11, 11]) == []                                           # one fast engine: ambiguous
    assert fast([50, 50, 50, 50, 40, 12]) == []                                   # five: ambiguous


def test_info_matches_dump_metadata(host_index, s10_dump):
    meta = dict(l.strip().split("=") for l in open(s10_dump + ".metadata.txt"))
    assert host_index.k() == int(meta["k"]) == 31
    assert host_index.num_colors() == int(meta["num_colors"]) == 10
    assert host_index.num_color_sets() == int(meta["num_color_sets"])
    assert host_index.num_unitigs() == int(meta["num_unitigs"])
    assert host_index.num_kmers() == int(meta["num_kmers"]) == 6898179


def test_hybrid_encoder_is_bit_identical_to_oracle(host_index, s10_oracle):
    ex = host_index.export()
    words, offs = s10_oracle.encoded_colors()
    assert np.array_equal(ex["color_offsets"], offs)
    assert np.array_equal(ex["color_words"], words)
    assert ex["thresholds"].tolist() == [10, 2, 7]  # u32(0.25*n), u32(0.75*n): hybrid.hpp:20-21


def test_dictionary_selfcheck(host_index):
    host_index.selfcheck(unitig_stride=7)  # every k-mer of every 7th unitig, both strands


def test_dump_and_container_agree(host_index, s10_dump):
    a = fulgor_amd.Index(s10_dump, device=-1).export()
    b = host_index.export()
    for key in ("unitig_bases", "unitig_off", "unitig_csid", "color_words", "color_offsets", "thresholds"):
        assert np.array_equal(a[key], b[key]), key


def test_export_unitigs_match_dump_text(host_index, s10_dump):
    ex = host_index.export()
    with open(s10_dump + ".unitigs.fa") as f:
        for u in range(50):
            hdr, seq = f.readline(), f.readline().strip()
            assert int(hdr.split("color_set_id=")[1]) == ex["unitig_csid"][u]
            a, b = int(ex["unitig_off"][u]), int(ex["unitig_off"][u + 1])
            assert bytes(ex["unitig_bases"][a:b]).decode() == seq


def test_queries_fail_loudly_on_host_only_handle(host_index):
    with pytest.raises(RuntimeError, match="no CPU path"):
        host_index.fetch_color_set_ids("ACGT" * 40)
    with pytest.raises(RuntimeError, match="no CPU path"):
        host_index.pseudoalign_threshold_union("ACGT" * 40, 0.8)
    with pytest.raises(RuntimeError, match="no CPU path"):
        host_index.pseudoalign_full_intersection([0, 1])


@pytest.mark.skipif(has_gpu(), reason="checks the no-GPU failure mode")
def test_open_on_device_fails_without_gpu(s10_fgidx):
    with pytest.raises(RuntimeError, match="no HIP device"):
        fulgor_amd.Index(s10_fgidx, device=0)


def test_reference_binary_index_stops_at_the_sshash_section(tmp_path):
    """a file written by the reference carries an sshash::dictionary behind the version number: its layout is not available
    here, the loader says so instead of guessing"""
    p = tmp_path / "x.mdfur"
    p.write_bytes(b"\x04\x02\x00" + bytes(range(64)))
    with pytest.raises(RuntimeError, match="fulgor dump"):
        fulgor_amd.Index(str(p), device=-1)
    q = tmp_path / "y.fur"
    q.write_bytes(b"\x03\x00\x00" + bytes(64))
    with pytest.raises(RuntimeError, match="MAJOR index version"):  # include/util.hpp:91-95
        fulgor_amd.Index(str(q), device=-1)


@pytest.mark.parametrize("suffix,index_type,psize,csize", [("fur", 0, 0, 0), ("mfur", 2, 3, 1), ("dfur", 1, 10, 4), ("mdfur", 3, 4, 2)])
def test_fur_layout_roundtrip(s10_fgidx, s10_dump, tmp_path, suffix, index_type, psize, csize):
    """f1, the unblocked half: the Fulgor-owned sections in the reference's visit order (version, [k2u], u2c + rank9, the
    colour-set container of the suffix's codec, filenames) written and parsed back. The k2u section is the engine's own
    block (no SSHash), the primitive layouts are from memory of upstream: a round trip of OUR files, not .fur support."""
    import filecmp
    ix = fulgor_amd.Index(s10_fgidx, device=-1)
    if index_type:
        ix.convert(index_type, psize, csize)  # fixes the partition / cluster shape that is written
    p = str(tmp_path / ("x." + suffix))
    ix.save(p)
    iy = fulgor_amd.Index(p, device=-1)
    assert iy.index_type == index_type
    a, b = ix.export(), iy.export()
    for key in ("unitig_bases", "unitig_off", "unitig_csid", "color_words", "color_offsets", "thresholds"):
        assert np.array_equal(a[key], b[key]), key
    assert a["k"] == b["k"] and a["color_bits"] == b["color_bits"]
    iy.selfcheck(unitig_stride=101)
    base = str(tmp_path / "dump")
    iy.dump(base)
    for sfx in (".metadata.txt", ".filenames.txt", ".unitigs.fa", ".color_sets.txt"):
        assert filecmp.cmp(base + sfx, s10_dump + sfx, shallow=False), sfx
    # the sections sit where the reference's visit order puts them: version first, filenames last
    raw = open(p, "rb").read()
    assert raw[:3] == b"\x04\x02\x00" and raw[3:11] == b"FGK2U001"
    assert raw.endswith(b"SAL_HA8462AA.fasta.gz")
    with open(p, "r+b") as f:  # a damaged file fails cleanly
        f.truncate(len(raw) - 100)
    with pytest.raises(RuntimeError):
        fulgor_amd.Index(p, device=-1)


def test_bad_container_is_rejected(tmp_path):
    p = tmp_path / "x.fgidx"
    p.write_bytes(b"NOTANIDX" + b"\0" * 64)
    with pytest.raises(RuntimeError, match="magic"):
        fulgor_amd.Index(str(p), device=-1)
    # a container of an earlier build says so and tells what to do
    p.write_bytes(b"FGIDX008" + b"\0" * 64)
    with pytest.raises(RuntimeError, match="container version 008.*rebuild the index"):
        fulgor_amd.Index(str(p), device=-1)


def test_readgen_is_sliceable_and_seeded(built):
    from conftest import S10_GENOMES
    from fulgor_amd.reads import ReadGenerator
    g = ReadGenerator(S10_GENOMES[:2])
    b, o = g.generate(0, 70000, 150, 42)
    b2, _ = g.generate(65000, 3000, 150, 42)  # crosses the 65536-read shard boundary
    assert np.array_equal(b[65000 * 150:68000 * 150], b2)
    b3, _ = g.generate(0, 1000, 150, 43)
    assert not np.array_equal(b[:150000], b3)
    assert set(np.unique(b).tolist()) <= set(b"ACGT")
    assert o[-1] == 70000 * 150


@pytest.mark.parametrize("index_type,psize,csize", [(1, 10, 4), (2, 3, 1), (3, 4, 2)])
def test_codec_conversion_roundtrip_on_host(s10_fgidx, tmp_path, index_type, psize, csize):
    """fgpu_convert re-encodes every colour set (self-checked against the hybrid decode inside the call);
    the converted index survives the container and still refuses queries without a GPU"""
    ix = fulgor_amd.Index(s10_fgidx, device=-1).convert(index_type, psize, csize)
    p = tmp_path / "x.fgidx"
    ix.save(p)
    iy = fulgor_amd.Index(str(p), device=-1)
    assert iy.index_type == index_type and iy.num_color_sets() == ix.num_color_sets()
    with pytest.raises(RuntimeError, match="no CPU path"):
        iy.pseudoalign_full_intersection([0, 1])
    with pytest.raises(RuntimeError, match="unknown index type"):
        ix.convert(7)


def test_dump_writes_the_reference_interchange_files(host_index, s10_dump, tmp_path):
    """`fulgor dump` (src/index.cpp:59-120): the files written from the container are byte-identical to the dump the
    container was built from, and the CLI entry point produces the same"""
    import filecmp, subprocess, sys
    base = str(tmp_path / "again")
    host_index.dump(base)
    for suffix in (".metadata.txt", ".filenames.txt", ".unitigs.fa", ".color_sets.txt"):
        assert filecmp.cmp(base + suffix, s10_dump + suffix, shallow=False), suffix


def test_host_code_under_sanitizers(c256_dump, tmp_path):
    """memory / race check of the host side of the library (SURVEY: race detection): dump ingestion, the multi-threaded
    dictionary-table and packed-block builders, the container and .fur round trips and the three codec conversions on the
    256-colour collection, built with AddressSanitizer + UBSan and with ThreadSanitizer; no report, tables identical after a
    reload"""
    import shutil, subprocess
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    from conftest import ROOT
    src = os.path.join(ROOT, "tests", "host_sanitize.cpp")
    def one(tag, flags):
        exe = str(tmp_path / ("host_" + tag))
        r = subprocess.run(["g++", "-O1", "-g", "-std=c++17"] + flags + ["-I", os.path.join(ROOT, "fulgor_amd", "csrc"), "-I",
                            os.path.join(ROOT, "include"), src, "-o", exe, "-lz", "-ldl", "-pthread"], capture_output=True, text=True)
        if r.returncode != 0:
            return "skip", r.stderr[-200:]
        out = tmp_path / tag
        out.mkdir()
        r = subprocess.run([exe, c256_dump, str(out)], capture_output=True, text=True, timeout=900)
        return "ran", r

    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(2) as pool:  # the two builds side by side: they are most of the CPU suite's time
        results = list(pool.map(lambda a: one(*a), (("asan", ["-fsanitize=address,undefined"]), ("tsan", ["-fsanitize=thread"]))))
    for kind, r in results:
        if kind == "skip":
            pytest.skip("sanitizer build not available: " + r)
        assert r.returncode == 0 and r.stdout.strip().endswith("host ok"), (r.stdout[-500:], r.stderr[-2000:])
        assert "Sanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-2000:]


def test_dump_round_trip_is_byte_identical_for_any_thread_count(s4546small_dump, tmp_path):
    """round-5 review, item 4 (reduced size; profiles/dump_roundtrip.py does it at full size): the 4546-colour test index written as
    the reference's dump files (src/index.cpp:59-120: fgpu_dump formats strips of sets on all threads), ingested again — the colour
    sets file is mapped, cut at line starts and parsed / encoded by every thread into a stream of its own, the streams joined bit by
    bit — and saved: the container is the original, byte for byte, with 1, 3 and all threads; a dump of the re-ingested index is the
    first dump, byte for byte; malformed colour-set files are refused with the reference's messages."""
    import hashlib
    import shutil
    fg, base = s4546small_dump
    want = hashlib.sha256(open(fg, "rb").read()).hexdigest()
    code = ("import sys; sys.path.insert(0, %r); import fulgor_amd; ix = fulgor_amd.Index(sys.argv[1], device=-1); ix.save(sys.argv[2]); "
            "len(sys.argv) > 3 and ix.dump(sys.argv[3])" % ROOT)
    for threads in ("1", "3", "0"):
        out = str(tmp_path / ("again_%s.fgidx" % threads))
        extra = [str(tmp_path / "dump2")] if threads == "3" else []
        r = subprocess.run([sys.executable, "-c", code, base, out] + extra, env=dict(os.environ, FULGOR_INGEST_THREADS=threads),
                           capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        assert hashlib.sha256(open(out, "rb").read()).hexdigest() == want, "threads=%s" % threads
    for suffix in (".color_sets.txt", ".unitigs.fa", ".metadata.txt", ".filenames.txt"):
        assert open(str(tmp_path / "dump2") + suffix, "rb").read() == open(base + suffix, "rb").read(), suffix
    # malformed colour-set files (a private copy of the small files; the colour sets edited)
    bad = str(tmp_path / "bad")
    for suffix in (".unitigs.fa", ".metadata.txt", ".filenames.txt"):
        shutil.copy(base + suffix, bad + suffix)
    lines = open(base + ".color_sets.txt", "rb").read().split(b"\n")
    cases = {"short": lines[:len(lines) // 2],                                                    # fewer lines than num_color_sets
             "size": lines[:7] + [b"size=0 "] + lines[8:],
             "line is short": lines[:7] + [lines[7].rsplit(b" ", 1)[0]] + lines[8:],              # one colour missing
             "not increasing": lines[:7] + [b"size=2 5 5"] + lines[8:],
             "malformed": lines[:7] + [b"sz=3 1 2 3"] + lines[8:]}
    for msg, ls in cases.items():
        with open(bad + ".color_sets.txt", "wb") as f:
            f.write(b"\n".join(ls))
        with pytest.raises(RuntimeError, match=msg):
            fulgor_amd.Index(bad, device=-1)
