"""Independent pure-Python restatement of the reference hot path (tests only).

Second, independent restatement of Roco-scientist/NGS-Barcode-Count's per-read
path, used ONLY to cross-check oracle/oracle.c (the C oracle) on small inputs.
Unlike the C oracle it builds the very regex string the reference builds
(src/info.rs:263-267, 291-294, 298) and runs it through Python's `re`, whose
leftmost-first semantics for this fixed-length pattern equal the Rust crate's.

Citations are into the reference tree (src/...).
"""
import re

import numpy as np


def _lines(text):
    """str::lines(): split on '\n', drop a final empty piece, strip one trailing '\r'"""
    parts = text.split("\n")
    if parts and parts[-1] == "":
        parts.pop()
    return [p[:-1] if p.endswith("\r") else p for p in parts]


class SequenceFormat:
    """src/info.rs:176-310"""

    def __init__(self, text):
        # info.rs:218-222
        data = "".join(l for l in _lines(text) if not l.startswith("#"))
        self.format_string = ""
        self.regions_string = ""
        self.regex_string = ""
        self.constant_region_length = 0
        self.barcode_num = 0
        self.barcode_lengths = []
        self.sample_length_option = None
        self.random_barcode = False
        self.sample_barcode = False
        # info.rs:232
        for m in re.finditer(r"(?i)(\{\d+\})|(\[\d+\])|(\(\d+\))|N+|[ATGC]+", data, flags=re.ASCII):
            g = m.group(0)
            name = None
            if "[" in g:
                name = "sample"
                self.sample_barcode = True
            elif "{" in g:
                self.barcode_num += 1
                name = "barcode%d" % self.barcode_num
            elif "(" in g:
                name = "random"
                self.random_barcode = True
            if name is not None:
                digits = int(re.search(r"\d+", g).group(0))
                self.regex_string += "(?P<%s>.{%d})" % (name, digits)
                if name == "sample":
                    self.sample_length_option = digits
                    ch = "S"
                elif "barcode" in name:
                    self.barcode_lengths.append(digits)
                    ch = "B"
                else:
                    ch = "R"
                self.regions_string += ch * digits
                self.format_string += "N" * digits
            elif "N" in g:
                self.regex_string += "[AGCT]{%d}" % g.count("N")
                self.format_string += g
            else:
                self.regex_string += g.upper()
                self.format_string += g
                self.regions_string += "C" * len(g)
                self.constant_region_length += len(g)
        self.length = len(self.format_string)
        self.format_regex = re.compile(self.regex_string)  # duplicate names raise, like Regex::new


def max_seq_errors(sample_errors, sample_size, barcode_errors, barcode_sizes, constant_errors, constant_size):
    """MaxSeqErrors::new, src/info.rs:490-543 -> (constant, sample, [barcodes])"""
    if sample_size is not None:
        ms = sample_errors if sample_errors is not None else sample_size // 5
    else:
        ms = 0
    mb = [barcode_errors if barcode_errors is not None else b // 5 for b in barcode_sizes]
    mc = constant_errors if constant_errors is not None else constant_size // 5
    return mc, ms, mb


def fix_error(mismatch_seq, possible_seqs, mismatches):
    """src/parse.rs:553-593"""
    best_match = None
    best = mismatches + 1
    keep = True
    for true_seq in possible_seqs:
        mm = 0
        for pc, cc in zip(true_seq, mismatch_seq):
            if pc != cc and cc != "N" and pc != "N":
                mm += 1
            if mm > best:
                break
        if mm == best:
            keep = False
        if mm < best:
            keep = True
            best = mm
            best_match = true_seq
    return best_match if (keep and best_match is not None) else None


def low_quality(qual, min_average, regions, start):
    """src/parse.rs:331-375 with f32 arithmetic"""
    scores = []
    prev = "\0"
    qs = [(ord(ch) - 33) & 0xFF for ch in qual]
    for score, t in zip(qs[start:], regions):
        if t != prev:
            if scores:
                s = np.float32(0.0)
                for v in scores:
                    s = np.float32(s + np.float32(v))
                avg = np.float32(s / np.float32(len(scores)))
                if avg < np.float32(min_average):
                    return True
                scores = []
            prev = t
            if t != "C":
                scores = [score]
        else:
            if t != "C":
                scores.append(score)
    return False


class Parser:
    """SequenceParser + Results + SequenceErrors, src/parse.rs:15-163, src/info.rs:661-808"""

    NAMES = ["matched", "constant_region", "sample_barcode", "barcode", "duplicates", "low_quality"]

    def __init__(self, scheme_text, samples=None, counted=None, max_sample=None, max_barcode=None, max_constant=None,
                 min_quality=0.0):
        self.fmt = SequenceFormat(scheme_text)
        self.samples_hash = dict(samples or {})  # seq -> id
        self.sample_seqs = set(self.samples_hash)
        self.counted_seqs = [set(s) for s in counted] if counted else []
        self.mc, self.ms, self.mb = max_seq_errors(max_sample, self.fmt.sample_length_option, max_barcode,
                                                   self.fmt.barcode_lengths, max_constant,
                                                   self.fmt.constant_region_length)
        self.min_quality = min_quality
        self.counters = dict.fromkeys(self.NAMES, 0)
        # Results::new, info.rs:678-732
        self.random = self.fmt.random_barcode
        self.results = {}
        self.omitted = False
        if self.samples_hash:
            for s in self.samples_hash:
                self.results[s] = {}
        elif not self.fmt.sample_barcode:
            self.results["barcode"] = {}
        else:
            self.omitted = True

    def add_count(self, sample, rnd, tup):
        """info.rs:735-808"""
        if self.omitted and sample not in self.results:
            self.results[sample] = {}
        if not self.random:
            if sample in self.results:
                d = self.results[sample]
                d[tup] = d.get(tup, 0) + 1
            return True
        key = "barcode" if sample == "" else sample
        if key in self.results:
            d = self.results[key]
            if tup not in d:
                d[tup] = {rnd if rnd is not None else ""}
            else:
                r = rnd if rnd is not None else ""
                if r in d[tup]:
                    return False
                d[tup].add(r)
                return True
        else:
            self.results[sample] = {tup: {rnd if rnd is not None else ""}}
        return True

    def process(self, seq, qual):
        f = self.fmt
        # check_and_fix_consant_region, parse.rs:151-163
        if not f.format_regex.search(seq):
            L = len(f.format_string)
            if len(seq) < L:
                seq = ""  # undefined in the reference (usize underflow); see oracle.c
            else:
                wins = [seq[i:i + L] for i in range(len(seq) - L)]  # parse.rs:291-304 (exclusive end)
                best = fix_error(f.format_string, wins, self.mc)
                if best is not None:
                    seq = "".join(o if n == "N" else n for o, n in zip(best, f.format_string))
                else:
                    seq = ""
        m = f.format_regex.search(seq)
        if not m:
            self.counters["constant_region"] += 1
            return "constant_region"
        if self.min_quality > 0.0:
            if low_quality(qual, self.min_quality, f.regions_string, m.start()):
                self.counters["low_quality"] += 1
                return "low_quality"
        # SequenceMatchResult::new, parse.rs:439-524
        sample_err = False
        gd = m.groupdict()
        if "sample" in gd:
            s = gd["sample"]
            if not self.sample_seqs or s in self.sample_seqs:
                sample = s
            else:
                fx = fix_error(s, self.sample_seqs, self.ms)
                if fx is not None:
                    sample = fx
                else:
                    sample = ""
                    sample_err = True
        else:
            sample = "barcode"
        counted_err = False
        counted = []
        if not sample_err:
            for i in range(f.barcode_num):
                b = gd["barcode%d" % (i + 1)]
                if self.counted_seqs:
                    if b not in self.counted_seqs[i]:
                        fx = fix_error(b, self.counted_seqs[i], self.mb[i])
                        if fx is not None:
                            b = fx
                        else:
                            counted_err = True
                            break
                counted.append(b)
        rnd = gd.get("random")
        if sample_err:
            self.counters["sample_barcode"] += 1
            return "sample_barcode"
        if counted_err:
            self.counters["barcode"] += 1
            return "barcode"
        if self.add_count(sample, rnd, ",".join(counted)):
            self.counters["matched"] += 1
            return "matched"
        self.counters["duplicates"] += 1
        return "duplicates"

    def rows(self):
        out = []
        for s, d in self.results.items():
            for t, v in d.items():
                out.append((s, t, len(v) if self.random else v))
        return sorted(out)
