"""The grouped contexts walk of Cheetah's decode passes (density_amd/csrc/decode_passes.hip::cheetah_walk<NB>: speculate by reads, ONE ordered pass over the group,
verify, take back, go again) as restated in tools/walk_group_model.py must give the sequential walk's contexts, running context and table (cheetah.rs:72,81,90,
97-102) on descriptor streams built to collide: few hashes, many predicted quads, reads of never-written contexts."""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import walk_group_model as m


@pytest.mark.parametrize("G", [1, 2, 4])
@pytest.mark.parametrize("n_hashes,p_pred", [(5, 0.5), (40, 0.4), (3000, 0.32), (60000, 0.3)])
def test_grouped_walk_is_the_sequential_walk(G, n_hashes, p_pred):
    for seed in range(8):
        assert m.check(seed, G, groups=4, n_hashes=n_hashes, p_pred=p_pred) >= 1.0


def test_a_group_without_predicted_quads_takes_one_pass():
    import random
    rnd = random.Random(1)
    pred, h, hw = m.random_stream(rnd, 128, 50, 0.0)
    ctx, c, passes = m.grouped(pred, h, hw, 7, {}, 2, 64)
    assert passes == 1 and ctx[0] == 7 and ctx[1:] == h[:-1] and c == h[-1]


@pytest.mark.parametrize("lag,garbage", [(0, 0.0), (1, 0.0), (3, 0.0), (3, 0.5), (8, 1.0)])
@pytest.mark.parametrize("n_hashes,p_pred", [(5, 0.5), (40, 0.4), (3000, 0.32)])
def test_team_walk_is_the_sequential_walk_however_stale_its_guesses(lag, garbage, n_hashes, p_pred):
    """Round 6's team walk: a turn's speculative reads see the table as it was `lag` turns earlier — or noise —, lane 0's context arrives with the token where the
    quad in front was predicted, and a wrong read is answered by guessing its chain again.  The verification must not care: contexts, running context and table
    are the sequential walk's."""
    for seed in range(6):
        assert m.check_team(seed, 2, groups=6, n_hashes=n_hashes, p_pred=p_pred, lag=lag, garbage=garbage) >= 1.0
