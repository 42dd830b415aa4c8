"""The oracle run through the reference's own thread structure (1 reader + N workers over a mutex-guarded deque, one
mutex-guarded Results: src/main.rs:69-121, src/input.rs:115-148, src/parse.rs:53-86) -- the form bench.py's
cpu_baseline times -- must give exactly what one context gives over the same reads."""
import pytest

import oracle_lib
import workloads


@pytest.mark.parametrize("name,n_sets,n_mol", [("config3", (4, 60, 60, 60), None), ("config4", (3, 8, 8, 8), 3000)])
def test_reference_thread_structure_equals_single_context(name, n_sets, n_mol):
    w = workloads.make(name, n_sets=n_sets, n_molecules=n_mol)
    n = 30_000  # > the 10,000-record back-pressure bound of the queue
    seq, qual = w.synth.generate_host(0, n)
    one = workloads.oracle_for(w)
    one.process_batch(seq, qual, w.read_len, w.read_len)
    workers = [workloads.oracle_for(w) for _ in range(3)]
    shared = workloads.oracle_for(w)
    total = oracle_lib.run_reference_threads(workers, shared, seq, qual, w.read_len, w.read_len)
    assert total == one.counters
    assert sum(total.values()) == n
    assert shared.rows() == one.rows()
    if name == "config4":
        assert total["duplicates"] > 1000  # the shared set did collapse PCR duplicates across workers
