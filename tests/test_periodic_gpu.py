"""GPU run of the periodic form of the all-blank chain pass (blance_amd/csrc/k_period.h) -- the DEFAULT since round 4
(first device run: round 4's first GPU session, config 3 at full size 6.81 -> 3.79 ms per call, oracle digest;
hip.Planner(periodic=False), options.reserved[2] & 256 or BLANCE_PERIODIC=0 switch it off).  Strict: a wrong plan from
this path turns the suite red."""
import json
import os

import pytest

from blance_amd import hip, synth

pytestmark = pytest.mark.gpu


def _golden_digests():
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "config_digests.json")) as f:
        return json.load(f)


def test_periodic_all_blank_pass():
    """The periodic form of the all-blank chain pass (k_period.h): config 3 at its full size -- two periods of
    128 steps walked per region, 32,512 copied -- has the oracle's digest; reduced and ragged shapes, a periodic stretch
    that ends early, and random regular trees (mostly the ways out) equal the oracle."""
    import test_periodic_emulated as T
    want = _golden_digests()["config3"]
    pl = hip.Planner(device_id=0, periodic=True)
    got = pl.plan(synth.config_flat(3))
    assert (got.iterations, got.digest()) == (want["iterations"], want["digest"])
    T.check_shapes(pl)
    T.check_trees(pl, P=65536, zones=32)
    T.check_weights_and_gaps(pl, P=65536, N=4096)
    pl.close()
    os.environ["BLANCE_PERIODIC_CUT"] = "1000"
    try:
        pl = hip.Planner(device_id=0, periodic=True)
        T.check_shapes(pl, T.SHAPES[:4])
        pl.close()
    finally:
        del os.environ["BLANCE_PERIODIC_CUT"]
    T.check_wide_regions(lambda planes: hip.Planner(device_id=0, periodic=True, planes=planes), P=65536)
    pl = hip.Planner(device_id=0, chain_min_parts=1, periodic=True)
    T.check_random(pl, range(7000, 7200))
    pl.close()


def test_way_out_of_the_periodic_form():
    """BLANCE_PERIODIC=0 (parsed, not merely present) and hip.Planner(periodic=False) walk every step of the all-blank
    pass (k_pass_chain_planes over the whole chain) -- same digest at config 3's full size; BLANCE_PERIODIC=1 keeps it on."""
    want = _golden_digests()["config3"]
    fp = synth.config_flat(3)
    pl = hip.Planner(device_id=0, periodic=False)
    got = pl.plan(fp)
    assert (got.iterations, got.digest()) == (want["iterations"], want["digest"])
    walked_all = got.struct.blank_pass_ms
    pl.close()
    for env, slower in (("0", True), ("1", False)):
        os.environ["BLANCE_PERIODIC"] = env
        try:
            pl = hip.Planner(device_id=0)
            got = pl.plan(fp)
            got = pl.plan(fp)
            assert got.digest() == want["digest"]
            # the full walk is ~3.3 ms, the periodic form ~0.3 ms: the knob really selects the path
            assert (got.struct.blank_pass_ms > 0.5 * walked_all) == slower, (env, got.struct.blank_pass_ms, walked_all)
            pl.close()
        finally:
            del os.environ["BLANCE_PERIODIC"]
