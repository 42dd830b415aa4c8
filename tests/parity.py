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


def check_per_read(case, plan, outcomes, idx, discard):
    """outcomes/idx: per-read arrays from the engine; compares every read with the oracle"""
    o = oracle_for(case)
    exp = [o.process(s, q) for s, q in case["reads"]]
    got_counts = {}
    for i, e in enumerate(exp):
        assert int(outcomes[i]) == CODE[e], (i, e, int(outcomes[i]), case["reads"][i])
        if e == "matched":
            got_counts[int(idx[i])] = got_counts.get(int(idx[i]), 0) + 1
    assert decode_rows(plan, got_counts, discard) == o.rows()
    return o
