"""Independent Python restatement of the reference's writers (src/output.rs:74-576,
src/info.rs:811-904) -- tests only.  Produces, from a Results-like dict, the files the reference
would write: {file name: (header line, sorted data rows)} plus the (file, count) pairs of the stats
file.  Row order inside a file is HashMap order in the reference, so rows are compared sorted."""


def create_header(barcode_num):  # output.rs:184-196
    if barcode_num > 1:
        return ",".join("Barcode_%d" % (i + 1) for i in range(barcode_num))
    return "Barcode"


def convert_code(code, counted_hash):  # output.rs:591-599
    return ",".join(counted_hash[i][b] for i, b in enumerate(code.split(",")))


class Writer:
    def __init__(self, results, samples_hash, counted_hash, barcode_num, prefix, merge_output, enrich):
        self.results = results            # {sample key: {tuple: count}}
        self.samples_hash = samples_hash  # {seq: id} or {}
        self.counted_hash = counted_hash  # [ {seq: id} ] or []
        self.barcode_num = barcode_num
        self.prefix = prefix
        self.merge = merge_output
        self.enrich = enrich and barcode_num >= 2  # main.rs:22-25
        self.files = {}
        self.output_files, self.output_counts = [], []
        self.single, self.double = {}, {}
        self.compounds_written = set()
        self.merged_count = 0
        self.merge_rows = []

    def name(self, key):
        return self.samples_hash.get(key, "barcode") if self.samples_hash else key

    def order(self, keys):
        keys = list(keys)
        if self.samples_hash:
            keys.sort(key=lambda k: self.samples_hash.get(k, "barcode"))
        return keys

    def add_single(self, sample, s, count):  # info.rs:840-866
        parts = s.split(",")
        for i in range(len(parts)):
            key = ",".join(parts[x] if x == i else "" for x in range(len(parts)))
            if sample in self.single:
                self.single[sample][key] = self.single[sample].get(key, 0) + count

    def add_double(self, sample, s, count):  # info.rs:869-904
        parts = s.split(",")
        n = len(parts)
        for a in range(n - 1):
            for b in range(a + 1, n):
                key = ",".join(parts[x] if x in (a, b) else "" for x in range(n))
                if sample in self.double:
                    self.double[sample][key] = self.double[sample].get(key, 0) + count

    def add_counts(self, sample, samples, kind):  # output.rs:199-361
        src = {"full": self.results, "single": self.single, "double": self.double}[kind]
        rows = []
        for code, count in list(src[sample].items()):
            written = convert_code(code, self.counted_hash) if (kind == "full" and self.counted_hash) else code
            if self.merge and code not in self.compounds_written:
                self.compounds_written.add(code)
                self.merged_count += 1
                self.merge_rows.append(written + "".join("," + str(src[s].get(code, 0)) for s in samples))
            rows.append("%s,%d" % (written, count))
            if kind == "full" and self.enrich:
                self.add_single(sample, written, count)
                if self.barcode_num > 2:
                    self.add_double(sample, written, count)
        return rows

    def write_enriched(self, kind):  # output.rs:364-485
        src = self.single if kind == "single" else self.double
        samples = self.order(src.keys())
        desc = "Single" if kind == "single" else "Double"
        header = create_header(self.barcode_num)
        merged_header = header + "".join("," + self.name(s) for s in samples)
        for s in samples:
            fn = "%s_%s_counts.%s.csv" % (self.prefix, self.name(s), desc)
            rows = self.add_counts(s, samples, kind)
            self.files[fn] = (header + ",Count", sorted(rows))
            self.output_files.append(fn)
            self.output_counts.append(len(rows))
        if self.merge:
            fn = "%s_counts.all.%s.csv" % (self.prefix, desc)
            self.files[fn] = (merged_header, sorted(self.merge_rows))
            self.output_files.append(fn)
            self.output_counts.insert(len(self.output_counts) - len(samples), self.merged_count)
            self.merged_count, self.merge_rows = 0, []

    def write(self):  # output.rs:74-181
        samples = self.order(self.results.keys())
        if self.enrich:
            for s in samples:
                self.single[s], self.double[s] = {}, {}
        header = create_header(self.barcode_num)
        if self.merge and len(samples) == 1:
            self.merge = False
        merged_header = header + "".join("," + self.name(s) for s in samples)
        for s in samples:
            fn = "%s_%s_counts.csv" % (self.prefix, self.name(s))
            rows = self.add_counts(s, samples, "full")
            self.files[fn] = (header + ",Count", sorted(rows))
            self.output_files.append(fn)
            self.output_counts.append(len(rows))
        if self.merge:
            fn = self.prefix + "_counts.all.csv"
            self.files[fn] = (merged_header, sorted(self.merge_rows))
            self.output_files.append(fn)
            self.output_counts.insert(0, self.merged_count)  # output.rs:171: counts and names end up misaligned
            self.merged_count, self.merge_rows = 0, []
        if self.enrich:
            self.write_enriched("single")
            if self.barcode_num > 2:
                self.write_enriched("double")
        return self
