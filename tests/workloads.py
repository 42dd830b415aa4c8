"""The BASELINE.json / SURVEY.md 8(d) workloads as (plan, synthetic generator) pairs.
Used by bench.py, __graft_entry__.smoke() and the GPU parity tests."""
import ngs_barcode_count_amd as pkg

DEL_SCHEME = "[8]AGCTACGAATCG{8}TGGA{8}TGGA{8}ACTAGAT"
DEL_RANDOM_SCHEME = DEL_SCHEME + "(12)TAGA"
CRISPR_SCHEME = "TTGTGGAAAGGACGAAACACCG{20}GTTTTAGAGCTAGAAATAGCAAGTT"
SET_SEED = 0xB0C0DE


class Workload:
    pass


def make(name, n_sets=None, lib=None, n_molecules=None, read_len=100, zipf=False, geo_total=0, synth_overrides=None):
    """name: config2 | config3 | config5.  n_sets = (samples, refs per counted barcode...) overrides the
    BASELINE sizes (tests use smaller sets so the CPU oracle stays fast)."""
    w = Workload()
    w.name = name
    w.read_len = read_len
    if name in ("config2", "config3"):
        sizes = n_sets or (4, 1000, 1000, 1000)
        w.scheme = DEL_SCHEME
        plan = pkg.Plan(w.scheme, lib=lib)
        # samples far apart; counted barcodes only >= 2 apart so that ties occur
        w.samples = pkg.make_set(SET_SEED, sizes[0], 8, 3, lib=lib)
        w.counted = [pkg.make_set(SET_SEED + 1 + i, sizes[1 + i], 8, 2, lib=lib) for i in range(3)]
        for i, s in enumerate(w.samples):
            plan.add_sample(s, "sample_%d" % i)
        for b, refs in enumerate(w.counted):
            for i, s in enumerate(refs):
                plan.add_counted(b, s, "bb%d_%d" % (b + 1, i))
        if name == "config2":
            w.kwargs = dict(max_sample=0, max_barcode=0, max_constant=0)
            plan.set_max_errors(0, 0, 0)
            w.synth_args = dict(seed=2)
            w.min_quality = 0.0
        else:
            w.kwargs = dict(min_quality=20.0)
            plan.set_min_quality(20.0)
            w.synth_args = dict(seed=3, p_sub=0.01, p_n=0.001, p_lowq=0.05)
            w.min_quality = 20.0
    elif name == "config4":
        # DEL + 12-nt random barcode, PCR duplicates: reads are draws from n_molecules molecules
        sizes = n_sets or (4, 1000, 1000, 1000)
        w.scheme = DEL_RANDOM_SCHEME
        plan = pkg.Plan(w.scheme, lib=lib)
        w.samples = pkg.make_set(SET_SEED, sizes[0], 8, 3, lib=lib)
        w.counted = [pkg.make_set(SET_SEED + 1 + i, sizes[1 + i], 8, 2, lib=lib) for i in range(3)]
        for i, s in enumerate(w.samples):
            plan.add_sample(s, "sample_%d" % i)
        for b, refs in enumerate(w.counted):
            for i, s in enumerate(refs):
                plan.add_counted(b, s, "bb%d_%d" % (b + 1, i))
        w.kwargs = {}
        # geo_total (the job's read count): copies per molecule geometric with mean 2, scattered over the job -- the
        # SURVEY.md 8(d) model; without it reads are uniform draws from n_molecules molecules (Poisson copies)
        w.synth_args = dict(seed=4, p_sub=0.01, p_n=0.001, n_molecules=n_molecules or 200_000_000, geo_total=geo_total)
        w.min_quality = 0.0
    elif name == "config5":
        sizes = n_sets or (100000,)
        w.scheme = CRISPR_SCHEME
        plan = pkg.Plan(w.scheme, lib=lib)
        w.samples = None
        w.counted = [pkg.make_set(SET_SEED + 5, sizes[0], 20, 3, lib=lib)]
        for i, s in enumerate(w.counted[0]):
            plan.add_counted(0, s, "guide_%d" % i)
        w.kwargs = {}
        # zipf: the hot-spot variant of SURVEY.md 8(d) -- guide ranks drawn with P(k) ~ 1/k instead of uniformly
        w.synth_args = dict(seed=5, p_sub=0.01, p_n=0.001, zipf=zipf)
        w.min_quality = 0.0
    else:
        raise KeyError(name)
    if synth_overrides:  # e.g. heavier error rates than the BASELINE model (parity tests of the deep search tiers)
        w.synth_args.update(synth_overrides)
    w.plan = plan
    w.synth = pkg.Synth(plan, read_len=w.read_len, **w.synth_args)
    return w


def oracle_for(w):
    import oracle_lib
    samples = {s: "sample_%d" % i for i, s in enumerate(w.samples)} if w.samples else None
    return oracle_lib.Oracle(w.scheme, samples=samples, counted=w.counted, **w.kwargs)


def bytes_per_read(w, f_matched):
    """algorithmic bytes per read, SURVEY.md 8(d): R + (R if quality filter) + 8 * f_matched"""
    # (random-barcode mode: one 8-byte key insert instead of the 8-byte counter read-modify-write)
    return w.read_len * (2 if w.min_quality > 0 else 1) + 8.0 * f_matched
