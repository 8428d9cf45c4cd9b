"""oracle/naive_proxy.cpp -- the string-keyed timing proxy of the reference's Go code path -- must plan what
the id-based oracle plans before its time is quoted as a baseline."""
import numpy as np
import pytest

from blance_amd import synth
from oracle import loader, naive_loader


@pytest.mark.parametrize("cfg,P,N", [(2, 300, 20), (2, 64, 4), (3, 200, 256), (3, 150, 300)])
def test_naive_proxy_matches_oracle(cfg, P, N):
    lines, stats = naive_loader.run(cfg, P, N)
    fp = synth.config_flat(cfg, P=P, N=N)
    want = loader.plan(fp)
    assert stats["sweeps"] == want.iterations
    k = 2 if cfg == 2 else 3
    nodes = np.array(want.out_nodes[:k * P]).reshape(P, k)
    got = {}
    for ln in lines:
        name, prim, rep = ln.split("|")
        got[int(name)] = [int(prim)] + [int(x) for x in rep.split(",")]
    # partition ids of config_flat are the decimal names in order
    assert [got[i] for i in range(P)] == nodes.tolist()
