"""Synthetic BASELINE.json configs: the direct flat generator equals the
string-level route through the interning layer, and the oracle reproduces the
independent cross-check hashes of SURVEY.md App. G."""
import hashlib

import numpy as np
import pytest

from blance_amd import abi, synth


@pytest.mark.parametrize("cfg,P,N", [(1, None, None), (2, 512, 32), (3, 600, 256), (3, 300, 200)])
def test_flat_equals_interned(cfg, P, N):
    a = synth.config_flat(cfg, P, N)
    b = synth.case_to_flat(synth.config_case(cfg, P, N))
    assert a.scalars == b.scalars
    for k in abi.I32_FIELDS + abi.U8_FIELDS:
        assert np.array_equal(a.arrays[k], b.arrays[k]), k


def survey_hash(fp, res):
    M, P, off = fp.n_states, fp.n_parts, res.out_off
    lines = []
    for i in range(P):
        f = [",".join(str(x) for x in res.out_nodes[off[i * M + m]:off[i * M + m + 1]]) for m in range(M)]
        lines.append("%d|%s\n" % (i, "|".join(f)))
    return hashlib.sha256("".join(lines).encode()).hexdigest()


@pytest.mark.parametrize("cfg,P,N,iters,sha", [
    (2, None, None, 2, "490877389672066443ac1f54910c43e833133cbe003c3b44906b32fd8905ccf0"),
    (3, 1024, 4096, 3, "49e512bc077da55aa431f5bdec7f7fdaf30608df2718e092c7fbd4613b2ca391"),
    (3, 4096, 4096, 3, "cb19b088bfdc42620b447237fcfab00ea63e3d425ec336d993fd14a7bae6475f"),
])
def test_oracle_reproduces_survey_hashes(cfg, P, N, iters, sha):
    from oracle import loader
    fp = synth.config_flat(cfg, P, N)
    res = loader.plan(fp)
    assert res.iterations == iters and res.n_warnings == 0
    assert survey_hash(fp, res) == sha


def test_metric_bookkeeping():
    fp = synth.config_flat(3, P=1024, N=4096)
    assert synth.assignments(fp) == 1024 * 3
    # primary 65,576 B + replica 98,344 B per partition per sweep
    assert synth.algorithmic_bytes_per_sweep(fp) == 1024 * (65576 + 98344)
    fp2 = synth.config_flat(2)
    assert synth.algorithmic_bytes_per_sweep(fp2) == 65536 * 8272


@pytest.mark.parametrize("P,N", [(700, 256), (2048, 512)])
def test_named_weighted_flat_equals_interned(P, N):
    """Workload (b) of bench.py's general-regime block: the numpy builder equals the string-level route (scrambled
    non-numeric names, Zipf partition weights) field for field."""
    a = synth.config3_named_weighted_flat(P, N)
    b = synth.case_to_flat(synth.config3_named_weighted_case(P, N))
    assert a.scalars == b.scalars
    for k in abi.I32_FIELDS + abi.U8_FIELDS:
        assert np.array_equal(a.arrays[k], b.arrays[k]), k


def test_rotated_instances_of_config3_plan_to_the_same_ids():
    """bench.py --gpus N: rank r plans config 3 with nodesAll starting 512 r names further on.  For a rotation by whole zones
    (128 names) every per-node hierarchy table differs from the unrotated instance's, zones and racks stay aligned blocks of
    ids, and the plan -- as ids -- is the same, so every rank's digest can be held against the oracle's; a rotation that is
    not a multiple of the zone size gives another plan."""
    from oracle import loader
    base = synth.config_flat(3, P=8192, N=512)
    rot = synth.config_flat(3, P=8192, N=512, rotate=256)
    assert (base.arrays["node_leaf_pos"] != rot.arrays["node_leaf_pos"]).any()
    assert (base.arrays["vertex_parent"] != rot.arrays["vertex_parent"]).any()
    want = loader.plan(base).digest()
    assert loader.plan(rot).digest() == want
    assert loader.plan(synth.config_flat(3, P=8192, N=512, rotate=48)).digest() != want
