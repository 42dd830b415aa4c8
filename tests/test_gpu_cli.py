"""End-to-end drop-in check of the `barcode-count` command line (FASTQ ingest -> gfx950 engine ->
CSV / stats writers) against the oracle's counts pushed through an independent restatement of the
reference's writers (tests/pyref_output.py)."""
import gzip
import os
import re
import subprocess

import numpy as np
import pytest

import cases
import parity
import pyref_output

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "ngs-barcode-count_amd", "csrc", "barcode-count")


def write_inputs(tmp, c, gz=False, crlf=False):
    nl = "\r\n" if crlf else "\n"
    fq = os.path.join(tmp, "reads.fastq" + (".gz" if gz else ""))
    text = "".join("@read_%d some description%s%s%s+%s%s%s" % (i, nl, s, nl, nl, q, nl) for i, (s, q) in enumerate(c["reads"]))
    if gz:
        with gzip.open(fq, "wt", newline="") as f:
            f.write(text)
    else:
        with open(fq, "w", newline="") as f:
            f.write(text)
    scheme = os.path.join(tmp, "scheme.txt")
    open(scheme, "w").write("# test scheme\n" + c["scheme"] + "\n")
    args = ["-f", fq, "-q", scheme]
    if c.get("samples"):
        p = os.path.join(tmp, "samples.csv")
        open(p, "w").write("Barcode,Sample_ID\n" + "".join("%s,%s\n" % kv for kv in c["samples"].items()))
        args += ["-s", p]
    if c.get("counted"):
        p = os.path.join(tmp, "counted.csv")
        open(p, "w").write("Barcode,Barcode_ID,Barcode_Number\n" + "".join(
            "%s,bb%d_%s,%d\n" % (s, b + 1, s, b + 1) for b, refs in enumerate(c["counted"]) for s in refs))
        args += ["-c", p]
    kw = c.get("kwargs", {})
    for flag, key in (("--max-errors-sample", "max_sample"), ("--max-errors-counted-barcode", "max_barcode"),
                      ("--max-errors-constant", "max_constant"), ("--min-quality", "min_quality")):
        if kw.get(key) is not None:
            args += [flag, str(kw[key])]
    return args


def expected(c, prefix, merge, enrich):
    o = parity.oracle_for(c)
    for s, q in c["reads"]:
        o.process(s, q)
    results = {k: {} for k in o.sample_keys()}
    for s, t, n in o.rows():
        results.setdefault(s, {})[t] = n
    counted_hash = [{s: "bb%d_%s" % (b + 1, s) for s in refs} for b, refs in enumerate(c["counted"])] if c.get("counted") else []
    barcode_num = o.barcode_num
    w = pyref_output.Writer(results, dict(c["samples"] or {}), counted_hash, barcode_num, prefix, merge, enrich).write()
    return o, w


def read_csv(path):
    lines = open(path).read().split("\n")
    assert lines[-1] == ""
    return lines[0], sorted(lines[1:-1])


def canonical(header, rows, n_fixed):
    """Without a sample file the column order of a merged file is HashMap order in the reference
    (output.rs:77-84, no sort): compare with the sample columns sorted by name."""
    cols = header.split(",")
    order = list(range(n_fixed)) + sorted(range(n_fixed, len(cols)), key=lambda i: cols[i])
    pick = lambda line: ",".join(line.split(",")[i] for i in order)
    return pick(header), sorted(pick(r) for r in rows)


@pytest.mark.parametrize("name,merge,enrich,gz", [("del_mismatch_quality", True, True, False),
                                                   ("del_random", True, False, True),
                                                   ("nosample", False, True, False),
                                                   ("crispr", True, True, False),
                                                   ("example_files_random_nosample", True, True, False),
                                                   ("nosample_with_sample_file", False, False, False),
                                                   ("raw_counted", False, True, False),
                                                   ("raw_sample", True, True, False),
                                                   ("raw_all_random", True, False, True),
                                                   ("example_files", True, True, False)])
def test_cli_outputs(tmp_path, name, merge, enrich, gz):
    c = cases.build_case(name, seed=31, n=2500)
    tmp = str(tmp_path)
    args = write_inputs(tmp, c, gz=gz)
    out = os.path.join(tmp, "out")
    os.makedirs(out)
    cmd = [CLI] + args + ["-o", out, "-p", "run1"] + (["-m"] if merge else []) + (["-e"] if enrich else [])
    res = subprocess.run(cmd, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr + res.stdout
    o, w = expected(c, "run1", merge, enrich)
    produced = sorted(f for f in os.listdir(out) if f.endswith(".csv"))
    assert produced == sorted(w.files), (produced, sorted(w.files))
    for fn, (header, rows) in w.files.items():
        h, r = read_csv(os.path.join(out, fn))
        if ".all." in fn:
            assert canonical(h, r, o.barcode_num) == canonical(header, rows, o.barcode_num), fn
        else:
            assert h == header, fn
            assert r == rows, fn
    # stdout / stats: counters, total sequences (gz counts one more, input.rs:69-73), files and counts
    total = len(c["reads"]) + (1 if gz else 0)
    assert ("Total sequences:             {:,}".format(total)) in res.stdout
    stats = open(os.path.join(out, "run1_barcode_stats.txt")).read()
    for label, key in (("Correctly matched sequences: ", "matched"), ("Constant region mismatches:  ", "constant_region"),
                       ("Sample barcode mismatches:   ", "sample_barcode"), ("Counted barcode mismatches:  ", "barcode"),
                       ("Duplicates:                  ", "duplicates"), ("Low quality barcodes:        ", "low_quality")):
        line = label + "{:,}".format(o.counters[key])
        assert line in res.stdout and line in stats, line
    listed = re.findall(r"File & barcodes counted: (\S+)\t([\d,]+)", stats)
    if c.get("samples"):
        assert [f for f, _ in listed] == w.output_files
        assert [int(n.replace(",", "")) for _, n in listed] == w.output_counts
    else:  # sample order is HashMap order without a sample file: same files, same counts, any order
        assert sorted(f for f, _ in listed) == sorted(w.output_files)
        assert sorted(int(n.replace(",", "")) for _, n in listed) == sorted(w.output_counts)
    assert stats.startswith("-TIME INFORMATION-\nStart: ") and "-FORMAT-\n" + o.format_string in stats
    assert "-BARCODE INFO-\nConstant region size: %d\n" % o.constant_region_length in stats


def test_cli_appends_stats_and_rejects_bad_input(tmp_path):
    c = cases.build_case("del_exact", seed=32, n=300)
    tmp = str(tmp_path)
    args = write_inputs(tmp, c, crlf=True)  # CRLF line ends are stripped on the plain path (input.rs:44)
    for _ in range(2):
        res = subprocess.run([CLI] + args + ["-o", tmp, "-p", "p"], capture_output=True, text=True)
        assert res.returncode == 0, res.stderr
    assert open(os.path.join(tmp, "p_barcode_stats.txt")).read().count("-TIME INFORMATION-") == 2
    o, w = expected(c, "p", False, False)
    for fn, (header, rows) in w.files.items():
        assert read_csv(os.path.join(tmp, fn)) == (header, rows)
    bad = os.path.join(tmp, "reads.txt")
    open(bad, "w").write("x")
    res = subprocess.run([CLI, "-f", bad, "-q", args[3]] + args[4:], capture_output=True, text=True)
    assert res.returncode != 0 and "only works with *.fastq" in res.stderr
    fq = os.path.join(tmp, "swapped.fastq")
    open(fq, "w").write("ACGTACGTACGT\n@name\n+\nIIIIIIIIIIII\n")
    res = subprocess.run([CLI, "-f", fq, "-q", args[3]] + args[4:], capture_output=True, text=True)
    assert res.returncode != 0 and "first line within the FASTQ contains DNA" in res.stderr


def test_count_fastq_through_the_abi(tmp_path):
    import ngs_barcode_count_amd as pkg
    from test_gpu_parity import make_plan
    c = cases.build_case("fmtn", seed=33, n=1500)  # ragged read lengths
    args = write_inputs(str(tmp_path), c)
    plan = make_plan(c)
    eng = pkg.Engine(plan, device=0)
    assert eng.count_fastq(args[1]) == len(c["reads"])
    o = parity.oracle_for(c)
    for s, q in c["reads"]:
        o.process(s, q)
    got = eng.counters()
    assert {k: got[k] for k in o.counters} == o.counters and eng.result_rows() == o.rows()
    # a trailing partial record is counted but never processed
    with open(args[1], "a") as f:
        f.write("@partial\nACGT\n")
    e2 = pkg.Engine(plan, device=0)
    assert e2.count_fastq(args[1]) == len(c["reads"]) + 1
    assert e2.counters()["total_reads"] == len(c["reads"])
    eng.close()
    e2.close()


def _count_file(plan, path):
    import ngs_barcode_count_amd as pkg
    eng = pkg.Engine(plan, device=0)
    total = eng.count_fastq(path)
    got, rows = eng.counters(), eng.result_rows()
    eng.close()
    return total, got, rows


@pytest.mark.parametrize("chunk", [4096, 4112, 65536])
@pytest.mark.parametrize("name", ["del_mismatch_quality", "fmtn"])
def test_records_straddling_ingest_chunks(tmp_path, monkeypatch, chunk, name):
    """the device-side framing: chunks are arbitrary byte ranges of the file, so with 4 KiB chunks nearly every chunk
    boundary falls inside a record; headers of varying length move the boundaries around"""
    from test_gpu_parity import make_plan
    monkeypatch.setenv("BC_INGEST_CHUNK", str(chunk))
    c = cases.build_case(name, seed=51, n=1200)
    fq = os.path.join(str(tmp_path), "reads.fastq")
    with open(fq, "w") as f:
        for i, (s, q) in enumerate(c["reads"]):
            f.write("@r%d %s\n%s\n+%s\n%s\n" % (i, "x" * (i % 37), s, "same" * (i % 3), q))
    o = parity.oracle_for(c)
    for s, q in c["reads"]:
        o.process(s, q)
    total, got, rows = _count_file(make_plan(c), fq)
    assert total == len(c["reads"]) and got["total_reads"] == len(c["reads"])
    assert {k: got[k] for k in o.counters} == o.counters and rows == o.rows()


def test_file_sizes_around_a_chunk_multiple_and_missing_final_newline(tmp_path, monkeypatch):
    """end-of-file accounting (input.rs:44, 86, 128-130): a last record without its final newline is a whole record on
    the plain path; file lengths just below, at and just above a multiple of the chunk size"""
    from test_gpu_parity import make_plan
    monkeypatch.setenv("BC_INGEST_CHUNK", "4096")
    c = cases.build_case("del_exact", seed=52, n=400)
    plan = make_plan(c)
    text = "".join("@r%d\n%s\n+\n%s\n" % (i, s, q) for i, (s, q) in enumerate(c["reads"]))
    o = parity.oracle_for(c)
    for s, q in c["reads"]:
        o.process(s, q)
    for variant, cut in (("exact", 0), ("no_final_newline", 1)):
        body = text[:len(text) - cut]
        for pad in (-1, 0, 1):
            # pad the FIRST header so that the file is `pad` bytes off a multiple of the chunk size
            want = ((len(body) + 4095) // 4096) * 4096 + 4096 + pad
            t2 = "@" + "h" * (want - len(body)) + body[1:]
            assert len(t2) == want + 0
            fq = os.path.join(str(tmp_path), "%s_%d.fastq" % (variant, pad + 1))
            open(fq, "w").write(t2)
            total, got, rows = _count_file(plan, fq)
            assert total == len(c["reads"]), (variant, pad, total)
            assert {k: got[k] for k in o.counters} == o.counters and rows == o.rows(), (variant, pad)


def test_quality_line_of_another_length_than_the_sequence_line(tmp_path):
    """trimmed or damaged FASTQ files have records whose quality line is shorter (or longer) than the
    sequence line; the reference zips scores with regions (parse.rs:340-345), the engine used to refuse the file"""
    from test_gpu_parity import make_plan
    rng = np.random.default_rng(77)
    c = cases.build_case("del_mismatch_quality", seed=53, n=1500)
    reads = []
    for s, q in c["reads"]:
        r = rng.random()
        if r < 0.2:
            q = q[:int(rng.integers(0, len(q)))]          # truncated quality line
        elif r < 0.3:
            q = q + "I" * int(rng.integers(1, 30))         # longer than the sequence line
        reads.append((s, q))
    c["reads"] = reads
    fq = os.path.join(str(tmp_path), "reads.fastq")
    with open(fq, "w") as f:
        for i, (s, q) in enumerate(reads):
            f.write("@r%d\n%s\n+\n%s\n" % (i, s, q))
    o = parity.oracle_for(c)
    for s, q in reads:
        o.process(s, q)
    total, got, rows = _count_file(make_plan(c), fq)
    assert total == len(reads)
    assert {k: got[k] for k in o.counters} == o.counters and rows == o.rows()
    assert o.counters["low_quality"] > 0 and o.counters["matched"] > 0


@pytest.mark.parametrize("name,gpus,gz,qual_at", [("del_mismatch_quality", 2, False, True), ("del_random", 3, False, False),
                                                  ("raw_counted", 2, False, False), ("crispr", 3, False, True),
                                                  ("del_mismatch_quality", 2, True, False)])
def test_cli_on_several_ranks_writes_what_one_rank_writes(tmp_path, name, gpus, gz, qual_at):
    """`barcode-count --gpus N`: a launcher that never touches the GPU, N rank processes (here all on device 0, joined by
    the message-file transport: the box has one GPU), every rank its share of the file's records, one exchange at the end
    (bc_fastq_count_shard + bc_engine_finish_all).  Same files, rows, counters and totals as the oracle -- with quality
    lines that begin with '@' where a shard boundary could mistake them for a header, and for a .gz input (which only the
    first rank can read)"""
    c = cases.build_case(name, seed=41, n=3001)
    if qual_at:  # quality lines starting with '@' (Phred 31): legal, and the classic trap of FASTQ splitting
        c["reads"] = [(s, ("@" + q[1:]) if q and i % 3 == 0 else q) for i, (s, q) in enumerate(c["reads"])]
    tmp = str(tmp_path)
    args = write_inputs(tmp, c, gz=gz)
    out = os.path.join(tmp, "out")
    os.makedirs(out)
    cmd = [CLI] + args + ["-o", out, "-p", "multi", "-m", "--gpus", str(gpus), "--devices", ",".join(["0"] * gpus), "--comm", "host"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=dict(os.environ, BC_INGEST_CHUNK="65536"))
    assert res.returncode == 0, res.stderr + res.stdout
    o, w = expected(c, "multi", True, False)
    produced = sorted(f for f in os.listdir(out) if f.endswith(".csv"))
    assert produced == sorted(w.files), (produced, sorted(w.files))
    for fn, (header, rows) in w.files.items():
        h, r = read_csv(os.path.join(out, fn))
        if ".all." in fn:
            assert canonical(h, r, o.barcode_num) == canonical(header, rows, o.barcode_num), fn
        else:
            assert (h, r) == (header, rows), fn
    total = len(c["reads"]) + (1 if gz else 0)
    assert ("Total sequences:             {:,}".format(total)) in res.stdout
    for label, key in (("Correctly matched sequences: ", "matched"), ("Constant region mismatches:  ", "constant_region"),
                       ("Sample barcode mismatches:   ", "sample_barcode"), ("Counted barcode mismatches:  ", "barcode"),
                       ("Duplicates:                  ", "duplicates"), ("Low quality barcodes:        ", "low_quality")):
        assert label + "{:,}".format(o.counters[key]) in res.stdout, label
    assert res.stdout.count("-FORMAT-") == 1  # one voice: rank 0's


def test_fastq_shards_tile_the_file(tmp_path, monkeypatch):
    """bc_fastq_count_shard: for 1..7 shards the shards' totals and counters add up to the whole file's, whatever the
    shard boundaries fall on (mid-header, mid-quality line, on a quality line that begins with '@'); a file whose lines
    do not come in fours is refused by the shard that notices"""
    import ngs_barcode_count_amd as pkg
    c = cases.build_case("del_mismatch_quality", seed=43, n=1200)
    c["reads"] = [(s, ("@" + q[1:]) if i % 2 == 0 else q) for i, (s, q) in enumerate(c["reads"])]
    tmp = str(tmp_path)
    write_inputs(tmp, c)
    fq = os.path.join(tmp, "reads.fastq")
    monkeypatch.setenv("BC_INGEST_CHUNK", "32768")
    plan = pkg.Plan(c["scheme"])
    for s, i in c["samples"].items():
        plan.add_sample(s, i)
    for b, refs in enumerate(c["counted"]):
        for s in refs:
            plan.add_counted(b, s, s)
    kw = c.get("kwargs", {})
    plan.set_max_errors(kw.get("max_sample"), kw.get("max_barcode"), kw.get("max_constant"))
    plan.set_min_quality(kw.get("min_quality", 0.0))
    whole = pkg.Engine(plan, device=0)
    assert whole.count_fastq(fq) == len(c["reads"])
    ref_counters, ref_rows = whole.counters(), whole.result_rows()
    whole.close()
    for n_shards in (2, 3, 5, 7):
        eng = pkg.Engine(plan, device=0)
        totals = [eng.count_fastq(fq, shard=k, n_shards=n_shards) for k in range(n_shards)]
        assert sum(totals) == len(c["reads"]) and all(t > 0 for t in totals), totals
        assert eng.counters() == ref_counters and eng.result_rows() == ref_rows
        eng.close()
    # a record with a line missing in the first half: the first shard sees lines that do not come in fours
    lines = open(fq).read().split("\n")
    del lines[4 * 100 + 2]
    bad = os.path.join(tmp, "broken.fastq")
    open(bad, "w").write("\n".join(lines))
    eng = pkg.Engine(plan, device=0)
    with pytest.raises(pkg.BarcodeCountError) as err:
        eng.count_fastq(bad, shard=0, n_shards=2)
    assert "records of four" in str(err.value)
    eng.close()


def test_gz_without_a_final_newline(tmp_path):
    """a .gz whose last line has no newline used to be refused.  The reference's read_line hands that line
    over as it is and post() pops the record's last character whatever it is (input.rs:137): the last record is scored
    with a quality line one character short -- here a quality line whose one low score sits in that last character, so
    the verdict changes with it.  A stream that stops inside a record's second line only adds to the total."""
    from test_gpu_parity import make_plan
    c = cases.build_case("del_mismatch_quality", seed=61, n=500)
    plan = make_plan(c)
    reads = list(c["reads"])
    # the last read: a clean construct whose last scored barcode base is the read's last base, quality 2 only there
    s_last, q_last = next((s, q) for s, q in reads if parity.oracle_for(c).process(s, q) == "matched")
    reads[-1] = (s_last, q_last)
    text = "".join("@r%d\n%s\n+\n%s\n" % (i, s, q) for i, (s, q) in enumerate(reads))
    for variant in ("fourth_line", "second_line", "third_line", "third_line_open"):
        body = {"fourth_line": text[:-1], "second_line": text + "@tail\nACGTACGT",
                # three lines into a record: the gz loop's extra read("") lands on "line 4" and the reference posts the
                # partial record with an EMPTY quality line (input.rs:69-73, 137; parse.rs:258-265) -- scored like any read
                "third_line": text + "@tail\n" + s_last + "\n+\n", "third_line_open": text + "@tail\n" + s_last + "\n+"}[variant]
        fq = os.path.join(str(tmp_path), variant + ".fastq.gz")
        with gzip.open(fq, "wb") as f:
            f.write(body.encode())
        o = parity.oracle_for(c)
        for i, (s, q) in enumerate(reads):
            o.process(s, q[:-1] if (variant == "fourth_line" and i == len(reads) - 1) else q)
        extra = 1 if variant.startswith("third_line") else 0
        if extra:
            assert o.process(s_last, "") == "matched"  # an empty quality line passes the filter
        total, got, rows = _count_file(plan, fq)
        assert total == len(reads) + 1, (variant, total)  # fourth_line: the gz path's extra read("") (input.rs:69-73); else: the partial record
        assert got["total_reads"] == len(reads) + extra
        assert {k: got[k] for k in o.counters} == o.counters and rows == o.rows(), variant
