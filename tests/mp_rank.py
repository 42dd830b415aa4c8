"""One rank of a multi-process GPU test (tests/test_gpu_multirank.py): counts its shard of a seeded workload on device 0
and joins the job's end-of-run exchange through the C ABI (bc_comm_create_host + bc_engine_finish_all); the root writes
the job's counters and rows as JSON.
    python tests/mp_rank.py <case> <rank> <world> <comm-dir> <n-total> <root> <out.json>"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def make_case(case):
    import ngs_barcode_count_amd as pkg
    import workloads
    if case == "dense":
        return workloads.make("config3", n_sets=(4, 60, 60, 60))
    if case == "dense_big":  # 32 M tuples: with two-level counting on, the tables stay sparse enough for the exchange's bit-map form
        return workloads.make("config3", n_sets=(4, 200, 200, 200))
    if case == "dense_hot":  # few tuples, many reads: counts far above 255 (the overflow side list of the byte-packed exchange)
        return workloads.make("config3", n_sets=(2, 3, 3, 3))
    if case == "random":
        return workloads.make("config4", n_sets=(4, 40, 40, 40), n_molecules=30_000)
    if case in ("sparse", "sparse_random"):
        # no conversion files at all: sample and barcodes are kept as captured (README.md "Barcode-seq")
        w = workloads.Workload()
        w.name, w.read_len, w.min_quality = case, 100, 0.0
        scheme = "[6]AGCTACGAATCG{7}TGGA{5}ACTAGAT" + ("(9)TAGA" if case == "sparse_random" else "")
        w.scheme = scheme
        w.samples, w.counted, w.kwargs = None, [[], []], {}
        w.plan = pkg.Plan(scheme)
        # reads drawn from a small pool of molecules so that keys repeat within and across ranks
        w.synth = pkg.Synth(w.plan, seed=11, read_len=100, p_sub=0.004, p_n=0.001, n_molecules=2_000)
        return w
    raise KeyError(case)


def main():
    case, rank, world, cdir, n_total, root, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], int(sys.argv[5]), int(sys.argv[6]), sys.argv[7]
    import torch
    import ngs_barcode_count_amd as pkg
    from ngs_barcode_count_amd import distributed as bcdist
    w = make_case(case)
    first, count = bcdist.shard(n_total, rank, world)
    eng = pkg.Engine(w.plan, device=0)
    R = w.read_len
    if count:
        dseq = torch.empty(count * R, dtype=torch.uint8, device="cuda")
        dqual = torch.empty(count * R, dtype=torch.uint8, device="cuda")
        w.synth.generate_device(0, None, first, count, dseq.data_ptr(), dqual.data_ptr())
        torch.cuda.synchronize()
        eng.submit_device(dseq.data_ptr(), dqual.data_ptr() if w.min_quality > 0 else None, count, R, R)
    comm = pkg.Comm.host(cdir, rank, world)
    counters, n_rows = eng.finish_all(comm, root)
    if rank == root:
        rows = eng.result_rows()
        assert len(rows) == n_rows
        with open(out, "w") as f:
            json.dump({"counters": counters, "rows": rows}, f)
    else:
        assert n_rows == 0 and not any(counters.values())
    comm.barrier()
    comm.close()
    eng.close()


if __name__ == "__main__":
    main()
