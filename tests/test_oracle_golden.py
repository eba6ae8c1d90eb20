"""oracle/burst_oracle.c against the committed kernel vectors that tests/golden/make_golden.py recorded from the
reference's own aded_mat16 / aded_mat16L / reScoreM_mat16 (via oracle/_ref/libref_harness.so).  Runs anywhere."""
import os

import numpy as np

import oraclelib as ol

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_kernel_vectors():
    v = np.load(os.path.join(G, "kernel_vectors.npz"))
    lut = ol.score_lut(1)
    n = len(v["E"])
    assert n >= 150
    hit_lanes = 0
    for i in range(n):
        rows = v["rows"][v["rows_off"][i]:v["rows_off"][i + 1]].reshape(-1, 16)
        q = v["q"][v["q_off"][i]:v["q_off"][i + 1]]
        E = int(v["E"][i])
        ret, mins = ol.aded_clump(rows, q, E, lut)
        assert ret == int(v["ret"][i]) and np.array_equal(mins, v["mins"][i]), i
        if ret > E:
            continue
        for z in range(16):
            lane = rows[:, z].copy()
            if mins[z] == ret:   # BEST/ALLPATHS/CAPITALIST bound = the clump minimum (burst.c:4220-4227)
                ok, h = ol.rescore_lane(q, lane, ret, lut)
                assert ok
                assert (int(h["ed"]), int(h["gapQ"]), int(h["gapR"]), int(h["finalPos"])) == \
                    (int(v["mins"][i][z]), int(v["gapQ"][i][z]), int(v["gapR"][i][z]), int(v["finalPos"][i][z])), (i, z)
                assert h["score"].view(np.uint32) == v["score_bits"][i][z]
                hit_lanes += 1
            if mins[z] <= E:     # FORAGE bound = the budget (burst.c:4224)
                ok, h = ol.rescore_lane(q, lane, E, lut)
                assert ok
                assert (int(h["ed"]), int(h["gapQ"]), int(h["gapR"]), int(h["finalPos"])) == \
                    (int(v["mins_f"][i][z]), int(v["gapQ_f"][i][z]), int(v["gapR_f"][i][z]), int(v["finalPos_f"][i][z])), (i, z)
                assert h["score"].view(np.uint32) == v["score_bits_f"][i][z]
    assert hit_lanes > 300


def test_error_budget_known_values():
    # SURVEY.md section 8 a8, verified against the reference: @0.98 98->1, 99..104->2; @0.97 100->3, 292->9; @0.95 100->5, 320->16
    f = ol.oracle().orc_error_budget
    assert [f(0.98, n) for n in (98, 99, 104)] == [1, 2, 2]
    assert (f(0.97, 100), f(0.97, 292), f(0.95, 100), f(0.95, 320)) == (3, 9, 5, 16)
    assert f(0.01, 100) == 254
