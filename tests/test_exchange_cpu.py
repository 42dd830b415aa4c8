"""The multi-GPU job's end-of-run exchange (csrc/bc_exchange.hpp) on CPU ranks: the SAME slicing / byte-packing /
overflow / owner-partition / gather logic the engine runs on device buffers (csrc/bc_comm.hip), instantiated here over
host buffers (tests/emu/exchange_host.cpp) and run by 2, 3 and 5 processes over the message-file transport of
csrc/bc_comm.hpp.  Expected results are recomputed independently in numpy."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "ngs-barcode-count_amd", "csrc")
EXE = os.path.join(ROOT, "tests", "emu", "exchange_host")
SRC = os.path.join(ROOT, "tests", "emu", "exchange_host.cpp")
DEPS = [SRC, os.path.join(CSRC, "bc_exchange.hpp"), os.path.join(CSRC, "bc_comm.hpp")]

M64 = (1 << 64) - 1


def _build():
    if not os.path.exists(EXE) or any(os.path.getmtime(d) > os.path.getmtime(EXE) for d in DEPS):
        subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-Wall", "-o", EXE, SRC])
    return EXE


def _mix(x):
    x = np.asarray(x, dtype=np.uint64)
    with np.errstate(over="ignore"):
        x = x ^ (x >> np.uint64(30))
        x = x * np.uint64(0xBF58476D1CE4E5B9)
        x = x ^ (x >> np.uint64(27))
        x = x * np.uint64(0x94D049BB133111EB)
        x = x ^ (x >> np.uint64(31))
    return x


def _table(seed, r, n):
    with np.errstate(over="ignore"):
        h = _mix(np.uint64(seed) * np.uint64(1000003) + np.uint64(r) * np.uint64(7919) + np.arange(n, dtype=np.uint64))
    kind = (h & np.uint64(1023)).astype(np.int64)
    hi = (h >> np.uint64(10))
    out = np.zeros(n, dtype=np.uint64)
    small = (kind >= 700) & (kind < 1000)
    out[small] = (hi[small] & np.uint64(7)) + np.uint64(1)
    mid = (kind >= 1000) & (kind < 1020)
    out[mid] = np.uint64(200) + (hi[mid] & np.uint64(127))
    big = kind >= 1020
    out[big] = np.uint64(0xFFFFFF00) + (hi[big] & np.uint64(255))
    return out.astype(np.uint32)


def _run_ranks(tmp_path, world, args_of):
    exe = _build()
    d = tmp_path / "msgs"
    d.mkdir()
    procs = [subprocess.Popen([exe, str(d), str(r), str(world)] + args_of(r), stderr=subprocess.PIPE, stdout=subprocess.PIPE)
             for r in range(world)]
    said = ""
    for r, p in enumerate(procs):
        out, err = p.communicate(timeout=120)
        assert p.returncode == 0, (r, err.decode()[-500:])
        said += out.decode()
    assert os.listdir(d) == []  # every message was consumed
    return said


def _sparse_count(seed, r, n, c):
    """what exchange_host's "bits" mode makes of a rank's counts: mostly 0 / 1, the full value for one entry in 97"""
    i = np.arange(n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        keep = _mix(np.uint64(seed) ^ (i * np.uint64(31) + np.uint64(r))) % np.uint64(97) == 0
    return np.where(keep, c, (c != 0).astype(np.uint32)).astype(np.uint32)


@pytest.mark.parametrize("world,n,root,how", [(2, 4096, 0, ""), (3, 10_001, 2, ""), (5, 777, 1, ""), (2, 3, 1, ""), (4, 2, 0, ""),
                                              (2, 100_000, 1, "bits"), (3, 70_001, 0, "bits"), (5, 200_003, 4, "bits"),
                                              (3, 70_001, 2, "bits_dense"), (2, 130, 0, "bits")])
def test_tables_are_summed_onto_the_root(tmp_path, world, n, root, how):
    """table lengths that do and do not divide by 64 x world, slices that come out empty, counts on both sides of the
    byte limit, u32 sums that wrap; "bits": counts split into a first-occurrence bit map and a nearly empty table, which
    travel as bit-map slices + a side list; "bits_dense": the same split with tables too full for that on some ranks --
    all ranks then fall back to bytes (table + bit) together"""
    out = tmp_path / "sum.bin"
    said = _run_ranks(tmp_path, world, lambda r: ["tables", str(n), "42", str(root), str(out), how])
    assert ("form bits" in said) == (how == "bits"), said  # the form every rank agreed on
    if how == "bits":  # sums above 3 are rare with two or three ranks (two bit planes carry them), not with five
        assert ("form bits+planes" in said) == (world <= 3), said
    raw = np.fromfile(out, dtype=np.uint8)
    got = raw[: 4 * n].view(np.uint32)
    counters = raw[4 * n:].view(np.uint64)
    exp = np.zeros(n, dtype=np.uint32)
    with np.errstate(over="ignore"):
        for r in range(world):
            c = _table(42, r, n)
            exp = exp + (_sparse_count(42, r, n, c) if how == "bits" else c)  # u32 arithmetic: wraps like the engine's atomics
    assert np.array_equal(got, exp)
    assert counters.tolist() == [world * (world + 1) // 2, 10 * world, n * world]


def _key_owner(keys, world):
    with np.errstate(over="ignore"):
        x = keys * np.uint64(0x9E3779B97F4A7C15)
    x = x ^ (x >> np.uint64(32))
    return ((x >> np.uint64(7)) % np.uint64(world)).astype(np.int64)


@pytest.mark.parametrize("world,n,root,words", [(2, 5000, -1, 1), (3, 4001, -1, 1), (3, 1000, 1, 1), (4, 0, -1, 1),
                                                (3, 3000, -1, 4), (2, 800, 0, 3)])
def test_keys_reach_their_owner(tmp_path, world, n, root, words):
    """hashed ownership (random-barcode plans: duplicates across ranks meet on one rank) and everything-to-the-root
    (raw-key plans); the u32 travelling with each key stays with it"""
    out = tmp_path / "keys"
    # (words > 1: keys several u64 wide, as the plans with long raw captures have them -- every key must arrive whole,
    # which the ranks check themselves)
    _run_ranks(tmp_path, world, lambda r: ["keys", str(n), "7", str(root), str(out), str(words)])
    sent = []
    for r in range(world):
        i = np.arange(n, dtype=np.uint64)
        with np.errstate(over="ignore"):
            keys = _mix(np.uint64(7) + (i * np.uint64(3) + np.uint64(r)) % np.uint64(2 * n + 1))
        vals = (r * 1000 + (np.arange(n) % 7)).astype(np.uint32)
        sent.append((keys, vals))
    rec = np.dtype([("k", "<u8"), ("v", "<u4")])
    for r in range(world):
        got = np.fromfile(str(out) + ".%d" % r, dtype=rec)
        exp = []
        for keys, vals in sent:
            own = _key_owner(keys, world) if root < 0 else np.full(n, root)
            exp += list(zip(keys[own == r].tolist(), vals[own == r].tolist()))
        assert sorted(zip(got["k"].tolist(), got["v"].tolist())) == sorted(exp)
    if root < 0 and n:
        # the point of ownership: equal keys from different ranks ended up on the same rank
        allk = np.concatenate([k for k, _ in sent])
        assert len(np.unique(allk)) < len(allk)
