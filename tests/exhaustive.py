"""An exhaustive small domain: a proof over that domain instead of sampled seeds.

Scheme `[2]AC{2}G` (L = 7; regions SSCCBBC; three constant bases), one mismatch allowed in the constants, in the sample
barcode and in the counted barcode, --min-quality 20.  Reads: EVERY string over {A, C, G, N} of every length 0..max_len,
each with EVERY quality string over {'#' (Phred 2), 'I' (Phred 40)} of the same length.  The domain therefore holds every
anchor offset, every repair candidate (incl. the never-tested last window and ties between windows), reads shorter than the
format, 'N' on constants / in barcodes / everywhere, exact, corrected, ambiguous and failed barcodes, and every way a
quality run can pass or fail."""
import numpy as np

SCHEME = "[2]AC{2}G"
SAMPLES = {"AC": "s_ac", "GG": "s_gg", "CA": "s_ca"}
COUNTED = [["AA", "AC", "GA", "CG", "NG"]]          # 'AA'/'AC' tie for 'AN'/'AG'..., a reference holding an 'N'
KWARGS = dict(max_sample=1, max_barcode=1, max_constant=1, min_quality=20.0)
LETTERS = np.frombuffer(b"ACGN", dtype=np.uint8)
QUALS = np.frombuffer(b"#I", dtype=np.uint8)


def case():
    return {"scheme": SCHEME, "samples": SAMPLES, "counted": COUNTED, "kwargs": KWARGS}


def domain(length, stride):
    """every (read, quality) pair of that length -> (seq[n, stride], qual[n, stride]) uint8, n = 8^length"""
    n = 8 ** length
    i = np.arange(n, dtype=np.int64)
    seq = np.full((n, stride), ord("N"), dtype=np.uint8)
    qual = np.full((n, stride), ord("!"), dtype=np.uint8)
    r, q = i >> length, i & ((1 << length) - 1)   # high part: the read (base 4), low part: the quality bits
    for p in range(length):
        seq[:, p] = LETTERS[(r >> (2 * p)) & 3]
        qual[:, p] = QUALS[(q >> p) & 1]
    return seq, qual


def all_lengths(max_len, stride):
    """the whole domain as one batch with per-read lengths"""
    seqs, quals, lens = [], [], []
    for ln in range(max_len + 1):
        s, q = domain(ln, stride)
        seqs.append(s)
        quals.append(q)
        lens.append(np.full(s.shape[0], ln, dtype=np.uint16))
    return np.concatenate(seqs), np.concatenate(quals), np.concatenate(lens)
