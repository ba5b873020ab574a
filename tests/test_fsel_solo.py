"""GPU tier (-m gpu): the SOLO form of the selector (fsel_solo_kernel, csrc/fsel.hip: one workgroup per frame, lazy evaluation in the
sense of Minoux' accelerated greedy).

A batch takes it on its own from 33 frames on (HORIZON <= 10); the selector tests of the other files use smaller batches and therefore
run the frame kernel's teams.  Here the same test bodies are collected once more with AVM_FSEL_SOLO=1, which forces the solo form
wherever it is possible: ids bit-exact and in selection order against the FP64 oracle and against the binary128 selection, the
std::map equal-key rule, non-finite inputs, the edge cases.  A candidate the lazy rounds never score is PROVEN to lose its round, so
nothing may differ - not an id, not the order, and the fValues only by the rounding of the two evaluation forms.  Plus what is
specific: the choice rule, the two forms against each other on a large batch, and that any lazy_tau gives the same result."""
import numpy as np
import pytest

from helpers import buffers, rel, synth

from test_gpu_parity import (  # noqa: F401
    test_selector_bench_batch_matches_the_oracle,
    test_selector_edge_cases,
    test_selector_equal_upper_bounds_follow_the_std_map_rule,
    test_selector_headline_500_to_150,
    test_selector_ids_over_many_frames,
    test_selector_information_and_ids,
    test_selector_non_finite_inputs_match_the_oracle,
)
from test_fsel_truth import test_gpu_selects_what_the_binary128_selection_selects  # noqa: F401

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def solo_form(monkeypatch, selector):
    monkeypatch.setenv("AVM_FSEL_SOLO", "1")
    yield


def test_every_re_collected_test_ran_the_solo_form_where_it_can(selector):
    pr = synth.make_fsel(2, horizon=10, n_cand=60, n_used=0, max_features=20)
    selector.select_batch(pr)
    assert selector.ctx.last_fsel_form() == "solo"
    pr = synth.make_fsel(1, horizon=13, n_cand=60, n_used=0, max_features=20)  # 3 H = 39: single-precision diagonals in LDS, exact bounds from memory
    selector.select_batch(pr)
    assert selector.ctx.last_fsel_form() == "solo"
    pr = synth.make_fsel(1, horizon=5, n_cand=600, n_used=0, max_features=20)   # more than 512 candidates: one launch per round
    selector.select_batch(pr)
    assert selector.ctx.last_fsel_form() == "rounds"


def test_which_form_a_batch_takes(selector, monkeypatch):
    monkeypatch.delenv("AVM_FSEL_SOLO", raising=False)
    small = synth.make_fsel(8, horizon=5, n_cand=60, n_used=2, max_features=20)
    selector.select_batch(small)
    assert selector.ctx.last_fsel_form() == "teams"
    big = synth.make_fsel(48, horizon=5, n_cand=60, n_used=2, max_features=20)
    selector.select_batch(big)
    assert selector.ctx.last_fsel_form() == "solo"
    monkeypatch.setenv("AVM_FSEL_FRAME", "0")  # (a test that asks for the launch-per-round path gets it)
    selector.select_batch(big)
    assert selector.ctx.last_fsel_form() == "rounds"


def test_the_two_forms_select_the_same_on_a_large_batch(selector, oracle, monkeypatch):
    """96 bench-shaped frames (500 candidates -> 150, horizon 10): solo against the teams, ids and order identical in every frame, and 12 of
    them against the oracle."""
    pr = synth.make_fsel(96, first_id=1000)
    dev = pr.to_device("cuda:0")
    a = selector.select_batch(dev).to_host()
    assert selector.ctx.last_fsel_form() == "solo"
    monkeypatch.setenv("AVM_FSEL_SOLO", "0")
    b = selector.select_batch(dev).to_host()
    assert selector.ctx.last_fsel_form() == "teams"
    assert (a.a["n_selected"] == 150).all()
    assert np.array_equal(a.a["n_selected"], b.a["n_selected"]) and np.array_equal(a.a["selected_ids"], b.a["selected_ids"])
    assert rel(a.a["fvalues"], b.a["fvalues"]) < 1e-10  # (the DPP elimination here, the matrix cores there)
    sub = synth.make_fsel(12, first_id=1000)
    oo = buffers.FselOutArrays.alloc(12, sub.dims["max_features"])
    oracle.fsel_select(sub, oo, n_threads=8)
    assert np.array_equal(a.a["selected_ids"][:12], oo.a["selected_ids"])


@pytest.mark.parametrize("tau", ["0.0", "0.5", "0.95", "1.5", "1e9"])
def test_lazy_tau_cannot_change_a_result(selector, oracle, monkeypatch, tau):
    """tau = 0: every live candidate is scored in every round (the full evaluation); tau huge: the first pass scores nobody but the
    never-scored, and the check of the pick finds everybody who matters.  Same ids, same order, same fValues bit for bit."""
    pr = synth.make_fsel(3, horizon=10, n_cand=200, n_used=4, max_features=60)
    monkeypatch.setenv("AVM_FSEL_LAZY_TAU", "0.0")
    ref = selector.select_batch(pr)
    monkeypatch.setenv("AVM_FSEL_LAZY_TAU", tau)
    out = selector.select_batch(pr)
    assert selector.ctx.last_fsel_form() == "solo"
    assert np.array_equal(out.a["selected_ids"], ref.a["selected_ids"]) and np.array_equal(out.a["n_selected"], ref.a["n_selected"])
    for q in range(3):  # (fvalues beyond n_selected are not written)
        n = int(ref.a["n_selected"][q])
        assert n > 0 and np.array_equal(out.a["fvalues"][q, :n], ref.a["fvalues"][q, :n])
    oo = buffers.FselOutArrays.alloc(3, 60)
    oracle.fsel_select(pr, oo)
    assert np.array_equal(out.a["selected_ids"], oo.a["selected_ids"])


def test_more_frames_than_compute_units_and_ragged_candidate_counts(selector, oracle):
    """600 small frames in one call (the workgroups take frames p, p + grid, ...), candidate counts from 0 to the maximum, some frames with
    every feature slot already used (kappa = 0): n_selected and ids identical to the oracle's in every frame."""
    P = 600
    pr = synth.make_fsel(P, first_id=5000, horizon=5, n_cand=48, n_used=12, n_cloud=20, max_features=12)
    rng = np.random.default_rng(3)
    pr.a["n_cand"][:] = rng.integers(0, 49, P)
    pr.a["n_cand"][:5] = [0, 1, 2, 48, 47]
    pr.a["n_used"][:] = 3
    pr.a["n_used"][7:11] = [12, 12, 11, 0]   # (max_features 12: kappa = 0, 0, 1, 12)
    out = selector.select_batch(pr)
    assert selector.ctx.last_fsel_form() == "solo"
    oo = buffers.FselOutArrays.alloc(P, 12)
    oracle.fsel_select(pr, oo, n_threads=8)
    assert np.array_equal(out.a["n_selected"], oo.a["n_selected"])
    assert np.array_equal(out.a["selected_ids"], oo.a["selected_ids"])
    assert int(oo.a["n_selected"][0]) == 0 and int(oo.a["n_selected"][7]) == 0


def test_reference_horizon_batch_on_the_solo_form(selector, oracle, monkeypatch):
    """HORIZON = 13 (utility/state_defs.h:8, what the reference is compiled with): 64 frames, 300 candidates -> 100, on the solo form
    (39 x 39 position blocks; the LDS copy of the candidates' diagonals is single precision there): ids and order identical to the teams' in
    every frame and to the oracle's in eight of them."""
    monkeypatch.delenv("AVM_FSEL_SOLO", raising=False)
    pr = synth.make_fsel(64, first_id=3000, horizon=13, n_cand=300, n_used=10, n_cloud=100, max_features=110)
    dev = pr.to_device("cuda:0")
    a = selector.select_batch(dev).to_host()
    assert selector.ctx.last_fsel_form() == "solo"
    monkeypatch.setenv("AVM_FSEL_SOLO", "0")
    b = selector.select_batch(dev).to_host()
    assert selector.ctx.last_fsel_form() == "teams"
    assert (a.a["n_selected"] == 100).all()
    assert np.array_equal(a.a["n_selected"], b.a["n_selected"]) and np.array_equal(a.a["selected_ids"], b.a["selected_ids"])
    sub = synth.make_fsel(8, first_id=3000, horizon=13, n_cand=300, n_used=10, n_cloud=100, max_features=110)
    oo = buffers.FselOutArrays.alloc(8, 110)
    oracle.fsel_select(sub, oo, n_threads=8)
    assert np.array_equal(a.a["selected_ids"][:8], oo.a["selected_ids"])


def test_equal_key_rule_at_the_reference_horizon(selector, oracle):
    """The mirror-pair frames of test_selector_equal_upper_bounds_follow_the_std_map_rule at HORIZON 13: there the solo form keeps the
    candidates' diagonals in single precision for its estimates - mirror candidates still get equal estimates, are compared by their exact
    bounds from memory, and the std::map rule is applied as the oracle applies it."""
    from test_gpu_parity import _mirror_frame

    for seed in (0, 1, 2):
        pr = _mirror_frame(H=13, seed=seed)
        oo = buffers.FselOutArrays.alloc(1, pr.dims["max_features"])
        oracle.fsel_select(pr, oo)
        out = selector.select_batch(pr)
        assert selector.ctx.last_fsel_form() == "solo"
        assert np.array_equal(out.a["n_selected"], oo.a["n_selected"]) and np.array_equal(out.a["selected_ids"], oo.a["selected_ids"])
        ids = pr.a["cand_id"][0].tolist()
        order = [ids.index(s) for s in oo.a["selected_ids"][0, : oo.a["n_selected"][0]]]
        assert any(k + 1 in order and k in order and order.index(k + 1) < order.index(k) for k in range(0, 40, 2))  # (the rule was exercised)
