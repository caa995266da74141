"""Synthetic salmonella_4546-shaped workload (bench / test data, not part of the engine).

The real salmonella_4546 collection (README.md:144-160 of the reference) is a download that is not
available offline; csrc/tools/synth_s4546.cpp builds a seeded stand-in with the same shape
(k=31, 4546 colours, 43,788,757 k-mers, ~2.1M unitigs, ~0.85M colour sets, ~0.18 GB colour stream)
on top of the real salmonella_10 sequences. Everything measured on it is labelled synthetic."""
import os
import subprocess

import numpy as np

from . import _build

DESCRIPTION = ("SYNTHETIC salmonella_4546-shaped index (seed 4546: 4546 colours, 43.79M 31-mers, 2.13M unitigs, "
               "0.85M colour sets over real salmonella_10 sequence + random accessory contigs; the real "
               "salmonella_4546 is not available offline)")

BIN = os.path.join(_build.PKG, "synth_s4546")
SRC = os.path.join(_build.CSRC, "tools", "synth_s4546.cpp")


def build_tool():
    deps = [SRC] + [os.path.join(_build.CSRC, "host", f) for f in os.listdir(os.path.join(_build.CSRC, "host"))]
    if _build._newer(BIN, deps):
        _build._run(["g++", "-O3", "-std=c++17", "-pthread", SRC, "-o", BIN])
    return BIN


def ensure_s4546(data_dir, s10_genomes):
    """returns (path of the .fgidx, [accessory sequence as uint8 array]); generates them if missing"""
    fg = os.path.join(data_dir, "s4546syn.v9.fgidx")
    acc = os.path.join(data_dir, "s4546syn.accessory.txt")
    if not (os.path.exists(fg) and os.path.exists(acc)):
        os.makedirs(data_dir, exist_ok=True)
        _build.build_tools()
        base = os.path.join(data_dir, "s10")
        if not os.path.exists(base + ".unitigs.fa"):
            subprocess.run([_build.BIN_CCDBG, "31", base] + list(s10_genomes), check=True)
        build_tool()
        tmp = fg + ".tmp"
        subprocess.run([BIN, base, tmp, acc], check=True)
        os.replace(tmp, fg)
    return fg, [np.fromfile(acc, dtype=np.uint8)]


def ensure_s4546_small(data_dir, s10_genomes):
    """a small index with the same 4546 colours and list shapes (every 24th salmonella_10 unitig, 2.2 M k-mers): for tests
    that move the whole index through the reference's text dump. Returns (path of the .fgidx, [accessory sequence])."""
    fg = os.path.join(data_dir, "s4546small.v9.fgidx")
    acc = os.path.join(data_dir, "s4546small.accessory.txt")
    if not (os.path.exists(fg) and os.path.exists(acc)):
        os.makedirs(data_dir, exist_ok=True)
        _build.build_tools()
        base = os.path.join(data_dir, "s10")
        if not os.path.exists(base + ".unitigs.fa"):
            subprocess.run([_build.BIN_CCDBG, "31", base] + list(s10_genomes), check=True)
        build_tool()
        tmp = fg + ".tmp"
        subprocess.run([BIN, base, tmp, acc, "4546", "24", "2200000"], check=True)
        os.replace(tmp, fg)
    return fg, [np.fromfile(acc, dtype=np.uint8)]


DESCRIPTION_CORE = ("SYNTHETIC salmonella_4546-shaped index, CORE-HEAVY profile (seed 4546, profile 1: nine core loci in ten are "
                    "carried by all 4546 strains, pieces lose at most one small clade; more than 70 % of the mapped reads have "
                    "results of at least 3409 colours)")


def ensure_s4546_core(data_dir, s10_genomes):
    """the core-heavy profile of the same generator (synth_s4546.cpp, profile 1). Returns (path of the .fgidx, [accessory])."""
    fg = os.path.join(data_dir, "s4546core.v9.fgidx")
    acc = os.path.join(data_dir, "s4546core.accessory.txt")
    if not (os.path.exists(fg) and os.path.exists(acc)):
        os.makedirs(data_dir, exist_ok=True)
        _build.build_tools()
        base = os.path.join(data_dir, "s10")
        if not os.path.exists(base + ".unitigs.fa"):
            subprocess.run([_build.BIN_CCDBG, "31", base] + list(s10_genomes), check=True)
        build_tool()
        tmp = fg + ".tmp"
        subprocess.run([BIN, base, tmp, acc, "4546", "1", "43788757", "1"], check=True)
        os.replace(tmp, fg)
    return fg, [np.fromfile(acc, dtype=np.uint8)]
