"""GPU run of the OPT-IN periodic form of the all-blank chain pass (blance_amd/csrc/k_period.h, hip.Planner(periodic=True)).
The path was written after round 3's GPU budget was spent: it is exact on the emulated kernels (tests/test_periodic_emulated.py,
config 3 at full size included) and has never run on the device, it is OFF by default, and nothing the default planner
launches depends on it.  Hence the non-strict xfail: the suite records the first device run either way (XPASS = the
oracle's digests at full size) without an experimental option turning the parity suite red.  Runs last (file name)."""
import json
import os

import pytest

from blance_amd import hip, synth

pytestmark = pytest.mark.gpu


def _golden_digests():
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "config_digests.json")) as f:
        return json.load(f)


@pytest.mark.xfail(strict=False, reason="opt-in path, first run on the device (see the module docstring)")
def test_periodic_all_blank_pass():
    """The opt-in periodic form of the all-blank chain pass (k_period.h): config 3 at its full size -- two periods of
    128 steps walked per region, 32,512 copied -- has the oracle's digest; reduced and ragged shapes, a periodic stretch
    that ends early, and random regular trees (mostly the ways out) equal the oracle."""
    import os
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
