"""Shared comparison logic: per-read outcomes / dense indices from the engine (or its host
emulation) against the CPU oracle."""
import numpy as np

import oracle_lib

CODE = {"matched": 0, "constant_region": 1, "sample_barcode": 2, "barcode": 3, "duplicates": 4, "low_quality": 5}


def oracle_for(case):
    kw = case.get("kwargs", {})
    return oracle_lib.Oracle(case["scheme"], samples=case.get("samples"), counted=case.get("counted"), **kw)


def _base5_text(code, n):
    out = []
    for _ in range(n):
        out.append("ACTGN"[code % 5])
        code //= 5
    return "".join(out)


def decode_rows(plan, idx_counts, discard):
    """{tuple key: count} -> sorted [(sample, tuple, count)] with sequences as keys.  The key is the
    mixed-radix number over (sample, barcode 1..n): a reference index where a known set exists, the
    base-5 code of the raw capture where none does"""
    if discard:
        return []
    groups = []  # (is_sample, known sequences or None, capture length)
    if plan.sample_barcode:
        s = [x for x, _ in plan.samples()]
        groups.append((True, s or None, plan.sample_length_option))
    for i, bl in enumerate(plan.barcode_lengths):
        st = [x for x, _ in plan.counted(i)]
        groups.append((False, st or None, bl))
    out = []
    for di, cnt in idx_counts.items():
        di = int(di)
        sample, parts = "barcode", []
        for is_sample, known, ln in reversed(groups):
            if known:
                v = known[di % len(known)]
                di //= len(known)
            else:
                v = _base5_text(di % (5 ** ln), ln)
                di //= 5 ** ln
            if is_sample:
                sample = v
            else:
                parts.append(v)
        out.append((sample, ",".join(reversed(parts)), int(cnt)))
    return sorted(out)


def check_per_read(case, plan, outcomes, idx, discard, rspace=0):
    """outcomes/idx: per-read arrays from the engine; compares every read with the oracle.
    rspace > 0 (random-barcode plans): idx holds tuple_index * rspace + random code and a matched
    read adds one distinct key to its tuple"""
    o = oracle_for(case)
    exp = [o.process(s, q) for s, q in case["reads"]]
    got_counts = {}
    for i, e in enumerate(exp):
        assert int(outcomes[i]) == CODE[e], (i, e, int(outcomes[i]), case["reads"][i])
        if e == "matched":
            di = int(idx[i]) // rspace if rspace else int(idx[i])
            got_counts[di] = got_counts.get(di, 0) + 1
    assert decode_rows(plan, got_counts, discard) == o.rows()
    return o


def apply_set_semantics(outcomes, idx, rcode, rspace):
    """host emulation only: turns "passed every test" into matched / duplicate by replaying the
    reads in order against a set of (tuple, random barcode) keys; returns (outcomes, keys)"""
    seen = set()
    out = outcomes.copy()
    keys = np.zeros(len(out), dtype=np.uint64)
    for i in range(len(out)):
        k = int(idx[i]) * int(rspace) + int(rcode[i])
        keys[i] = k
        if out[i] == 0:
            if k in seen:
                out[i] = CODE["duplicates"]
            else:
                seen.add(k)
    return out, keys
