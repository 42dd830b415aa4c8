"""Small random-read generators for the parity tests (numpy RNG, python strings).

These are adversarial/edge-case inputs at sizes the CPU oracle and the pure-Python restatement
finish in seconds.  The big seeded workloads of BASELINE.json come from the product's own
counter-based generator (bc_synth_*), not from here."""
import re

import numpy as np

ACGT = "ACGT"


def rand_seq(rng, n, alphabet=ACGT):
    return "".join(alphabet[i] for i in rng.integers(0, len(alphabet), n))


def make_set(rng, n, k, min_dist=1):
    out = []
    seen = set()
    tries = 0
    while len(out) < n and tries < 100000:
        tries += 1
        s = rand_seq(rng, k)
        if s in seen:
            continue
        if min_dist > 1 and any(sum(a != b for a, b in zip(s, t)) < min_dist for t in out):
            continue
        seen.add(s)
        out.append(s)
    return out


def scheme_layout(scheme_text):
    """[(kind, value)] in order: kind in S,B,R (value=len), C (value=literal), N (value=count)"""
    data = "".join(l for l in scheme_text.split("\n") if not l.startswith("#"))
    out = []
    for m in re.finditer(r"(\{\d+\})|(\[\d+\])|(\(\d+\))|N+|[ATGC]+", data):
        g = m.group(0)
        if g[0] == "[":
            out.append(("S", int(g[1:-1])))
        elif g[0] == "{":
            out.append(("B", int(g[1:-1])))
        elif g[0] == "(":
            out.append(("R", int(g[1:-1])))
        elif g[0] == "N":
            out.append(("N", len(g)))
        else:
            out.append(("C", g))
    return out


def mutate(rng, s, p_sub, p_n, p_other=0.0):
    out = []
    for ch in s:
        r = rng.random()
        if r < p_sub:
            out.append(rng.choice([c for c in ACGT if c != ch]))
        elif r < p_sub + p_n:
            out.append("N")
        elif r < p_sub + p_n + p_other:
            out.append(rng.choice(list("acgtnRYX.")))
        else:
            out.append(ch)
    return "".join(out)


def gen_reads(rng, scheme_text, n, read_len, samples=None, counted=None, p_sub=0.02, p_n=0.003, p_other=0.0,
              p_garbage=0.03, p_lowq=0.1, dup_frac=0.0, var_len=False):
    """Returns list of (seq, qual).  Constructs are placed at a random offset (sometimes flush with
    the 3' end, sometimes the read is exactly the construct) so every repair quirk is exercised."""
    lay = scheme_layout(scheme_text)
    L = sum(v if k != "C" else len(v) for k, v in lay)
    reads = []
    pool = []
    for i in range(n):
        if pool and rng.random() < dup_frac:
            reads.append(pool[rng.integers(0, len(pool))])
            continue
        rl = read_len if not var_len else int(rng.integers(max(L - 3, 1), read_len + 1))
        if rng.random() < p_garbage or rl < L:
            seq = rand_seq(rng, rl)
        else:
            parts = []
            bi = 0
            for k, v in lay:
                if k == "C":
                    parts.append(v)
                elif k == "N":
                    parts.append(rand_seq(rng, v))
                elif k == "S":
                    parts.append(samples[rng.integers(0, len(samples))][:v].ljust(v, "A") if samples and rng.random() < 0.9
                                 else rand_seq(rng, v))
                elif k == "B":
                    ref = counted[bi] if counted else None
                    parts.append(ref[rng.integers(0, len(ref))][:v].ljust(v, "C") if ref and rng.random() < 0.9
                                 else rand_seq(rng, v))
                    bi += 1
                else:
                    parts.append(rand_seq(rng, v))
            construct = mutate(rng, "".join(parts), p_sub, p_n, p_other)
            slack = rl - L
            r = rng.random()
            if r < 0.1:
                off = slack  # flush with the 3' end: the window repair never tests
            elif r < 0.2:
                off = 0
            else:
                off = int(rng.integers(0, slack + 1))
            seq = rand_seq(rng, off) + construct + rand_seq(rng, slack - off)
        q = rng.integers(30, 41, len(seq))
        if rng.random() < p_lowq:
            a = int(rng.integers(0, max(len(seq) - 8, 1)))
            q[a:a + 10] = rng.integers(2, 16, len(q[a:a + 10]))
        qual = "".join(chr(33 + int(x)) for x in q)
        reads.append((seq, qual))
        pool.append((seq, qual))
    return reads


def to_arrays(reads, stride=None):
    """pack reads into fixed-stride uint8 arrays (+ per-read lengths); pad byte is '\\n'"""
    n = len(reads)
    mx = max(len(s) for s, _ in reads)
    stride = stride or mx
    seq = np.full((n, stride), 10, dtype=np.uint8)
    qual = np.full((n, stride), 10, dtype=np.uint8)
    lens = np.zeros(n, dtype=np.uint16)
    for i, (s, q) in enumerate(reads):
        seq[i, :len(s)] = np.frombuffer(s.encode(), dtype=np.uint8)
        qual[i, :len(q)] = np.frombuffer(q.encode(), dtype=np.uint8)
        lens[i] = len(s)
    return seq, qual, lens
