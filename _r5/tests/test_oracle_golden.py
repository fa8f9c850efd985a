"""The C oracle (oracle/wfst_oracle.c) must reproduce the reference-generated golden vectors
bit for bit: words, transition-ids, per-hop labels/costs, tot/lm score, surviving token and
link counts, per-frame token counts and best costs.  CPU only."""
import pytest

import pyoracle
from golden_util import GOLDEN_NAMES, Golden, check_result


@pytest.mark.parametrize("name", GOLDEN_NAMES)
def test_oracle_reproduces_golden(name, oracle, tmp_path):
    g = Golden(name)
    h = oracle.load_graph(g.write_graph(str(tmp_path / "g.bin")))
    n = 0
    for k, cd, md, ui in g.cases():
        trace = md.pop("trace", False)
        r = oracle.decode(h, pyoracle.Config(**cd), g.utts[ui], g.tid2pdf, trace=trace, **md)
        check_result(r, g.expected(k), "%s case %d" % (name, k))
        n += 1
    oracle.free_graph(h)
    assert n == len(g.meta["cases"]) and n > 0


def test_quirk_golden_is_the_documented_one():
    """SURVEY.md section 7 'Traceback quirk': of two parallel arcs 0->1 the reference returns the
    higher-index one (word 22) while its forward link is alive, and the arg-min one (word 11)
    only once PruneForwardLinks has excised the worse link (lattice_beam 0.25 + FinalizeDecoding)."""
    g = Golden("quirk_parallel_arcs")
    got = {}
    for k, cd, md, ui in g.cases():
        got[(cd["lattice_beam"], md.get("finalize", True))] = int(g.expected(k)["words"][0])
    assert got == {(8.0, True): 22, (8.0, False): 22, (0.25, True): 11, (0.25, False): 22}


def test_order_free_mode_coincides_with_the_reference_where_order_cannot_matter(oracle, tmp_path):
    """The oracle's order-free switch (what the GPU path is held to) against the
    reference-generated goldens: identical results for every beam-only case (no parallel arcs in
    hclg600 / eps_chains on the best path), and for most cases where max_active / min_active bind
    (there the reference's own cutoff depends on its visiting order)."""
    beam_only = binding = binding_same = 0
    for name in ("hclg600", "eps_chains", "no_final"):
        g = Golden(name)
        h = oracle.load_graph(g.write_graph(str(tmp_path / (name + ".bin"))))
        try:
            oracle.set_order_free(True)
            for k, cd, md, ui in g.cases():
                trace = md.pop("trace", False)
                r = oracle.decode(h, pyoracle.Config(**cd), g.utts[ui], g.tid2pdf, trace=trace, **md)
                e = g.expected(k)
                if cd["max_active"] >= 1000 and cd["min_active"] == 0:
                    check_result(r, e, "%s case %d (order-free)" % (name, k), check_counts=False)
                    beam_only += 1
                else:
                    binding += 1
                    binding_same += int(bool(r.ok) == bool(int(e["ok"])) and len(r.tids) == len(e["tids"]) and (r.tids == e["tids"]).all())
        finally:
            oracle.set_order_free(False)
            oracle.free_graph(h)
    assert beam_only >= 30 and binding >= 10 and binding_same >= binding // 2, (beam_only, binding, binding_same)
