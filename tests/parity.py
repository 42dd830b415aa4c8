"""Shared comparison logic: per-read outcomes / dense indices from the engine (or its host
emulation) against the CPU oracle."""
import numpy as np

import oracle_lib

CODE = {"matched": 0, "constant_region": 1, "sample_barcode": 2, "barcode": 3, "duplicates": 4, "low_quality": 5}


def oracle_for(case):
    kw = case.get("kwargs", {})
    return oracle_lib.Oracle(case["scheme"], samples=case.get("samples"), counted=case.get("counted"), **kw)


def decode_rows(plan, idx_counts, discard):
    """{dense index: count} -> sorted [(sample, tuple, count)] with the plan's sequences as keys"""
    if discard:
        return []
    samples = [s for s, _ in plan.samples()] if plan.sample_barcode else ["barcode"]
    sets = [[s for s, _ in plan.counted(i)] for i in range(plan.barcode_num)]
    out = []
    for di, cnt in idx_counts.items():
        di = int(di)
        parts = []
        for st in reversed(sets):
            parts.append(st[di % len(st)])
            di //= len(st)
        out.append((samples[di], ",".join(reversed(parts)), int(cnt)))
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
