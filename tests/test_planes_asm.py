"""The hand-written scalar loop of k_pass_chain_planes (planes_walk_w2, blance_amd/csrc/k_pass_chain.h) against the
definition of the plane automaton.  The SIMT emulator compiles the C++ twin of that loop (inline assembly does not exist
for it), so the assembly itself otherwise runs only in the device tests: here its text -- the strings of the
preprocessed translation unit, all eight instantiations K = 1..4 x {class masks read from lanes, class masks by
arithmetic} -- is executed by tests/gcn_scalar.py inside a Python rendering of the kernel's batch loop (the C++ around
the asm: resume after an exit, generic picks from any plane, dropping an empty lowest plane) and compared, pick for pick
and plane for plane, with the automaton's definition applied step by step (planes_pick_from)."""
import os
import random
import shutil

import pytest

from gcn_scalar import Machine, Program, preprocessed_asm_templates

K_PLANES = 4
M64 = (1 << 64) - 1

pytestmark = pytest.mark.skipif(not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")), reason="needs hipcc -E")


def pick_from(P, E, cls_mask, more):
    """planes_pick_from<2, 0, MORE>: lowest plane with a leaf outside E, its lowest such leaf (word 0 before word 1), moved
    one plane up; MORE: the leaf's class joins E.  Returns the leaf, -1 (no candidate) or -2 (it left the planes)."""
    for j in range(K_PLANES):
        m = [P[j][u] & ~E[u] & M64 for u in range(2)]
        if m[0] | m[1]:
            u = 0 if m[0] else 1
            b = (m[u] & -m[u]).bit_length() - 1
            P[j][u] ^= 1 << b
            if j + 1 < K_PLANES:
                P[j + 1][u] |= 1 << b
            if more:
                cm = cls_mask[64 * u + b]
                E[0] |= cm & M64
                E[1] |= cm >> 64
            return 64 * u + b if j + 1 < K_PLANES else -2
    return -1


def drop_empty_plane(P):
    if P[0][0] | P[0][1]:
        return 0
    for j in range(K_PLANES - 1):
        P[j] = P[j + 1]
    P[K_PLANES - 1] = [0, 0]
    return 1


def reference(K, P, ex, cls_mask, nb):
    P = [list(p) for p in P]
    picks, shifts, trouble = [[0] * 64 for _ in range(K)], 0, False
    for r in range(nb):
        E = [ex[r] & M64, ex[r] >> 64]
        for slot in range(K):
            f = pick_from(P, E, cls_mask, slot + 1 < K)
            trouble |= f < 0
            picks[slot][r] = f & 0xffffffff
        shifts += drop_empty_plane(P)
    return picks, P, shifts, trouble


def with_asm(prog, K, P, ex, cls_mask, nb, cls_m1, cls_ones):
    """k_pass_chain_planes' `while (r < nb)` loop with planes_walk_w2 interpreted from its assembly text."""
    P = [list(p) for p in P]
    lanes = lambda f: [f(l) & 0xffffffff for l in range(64)]
    ops = {"ex%d" % x: lanes(lambda l, x=x: ex[l] >> (32 * x)) for x in range(4)}
    for u in range(2):
        for x in range(4):
            ops["lm%d%d" % (u, x)] = lanes(lambda l, u=u, x=x: cls_mask[64 * u + l] >> (32 * x))
    for c in range(K):
        ops["w%d" % c] = [0] * 64
    ops.update(sm1=cls_m1, sones=cls_ones, nb=nb)
    shifts, trouble, r, exits, executed = 0, False, 0, 0, 0
    while r < nb:
        for j, nm in enumerate(("p0", "p1", "p2")):
            ops[nm + "l"], ops[nm + "h"] = P[j]
        ops.update(r=r, slot=0, elo=0, ehi=0)
        m = Machine(ops)
        m.run(prog)
        executed += m.executed
        for j, nm in enumerate(("p0", "p1", "p2")):
            P[j] = [ops[nm + "l"], ops[nm + "h"]]
        r = ops["r"]
        if r >= nb:
            break
        exits += 1
        E, slot0 = [ops["elo"], ops["ehi"]], ops["slot"]
        for slot in range(K):
            if slot >= slot0:
                f = pick_from(P, E, cls_mask, slot + 1 < K)
                trouble |= f < 0
                ops["w%d" % slot][r] = f & 0xffffffff
        shifts += drop_empty_plane(P)
        r += 1
    return [ops["w%d" % c] for c in range(K)], P, shifts, trouble, exits, executed


def random_case(rng, arith):
    n_leaves = rng.choice([128, 128, 96, 64, 70])
    if arith:
        S = rng.choice([4, 8, 16, 32, 64])
        cls_of = [l // S for l in range(128)]
        cls_m1, cls_ones = S - 1, (1 << S) - 1
    else:
        n_cls = rng.randint(3, 24)
        cls_of = [rng.randrange(n_cls) for _ in range(128)]          # classes scattered over both words
        cls_m1, cls_ones = 0, 1
    members = {}
    for l in range(n_leaves):
        members[cls_of[l]] = members.get(cls_of[l], 0) | (1 << l)
    cls_mask = [members.get(cls_of[l], 0) if l < n_leaves else 0 for l in range(128)]
    # counts within a few levels of each other: most leaves on the two lowest planes
    P = [[0, 0] for _ in range(K_PLANES)]
    weights = rng.choice([(8, 3, 1, 0), (1, 0, 0, 0), (1, 6, 1, 0), (3, 3, 3, 0), (0, 1, 0, 0), (1, 1, 0, 0)])
    for l in range(n_leaves):
        j = rng.choices(range(K_PLANES), weights)[0]
        P[j][l >> 6] |= 1 << (l & 63)
    while not (P[0][0] | P[0][1]):                                       # the kernel starts with its minimum on plane 0
        drop_empty_plane(P)
    nb = rng.choice([64, 64, 37, 1, 2, 5])
    ex = []
    for _ in range(64):
        top = rng.randrange(n_leaves)
        e = cls_mask[top]
        for _ in range(rng.choice([0, 0, 1, 2])):
            e |= 1 << rng.randrange(n_leaves)                            # a higher priority node of the step
        ex.append(e)
    return P, ex, cls_mask, nb, cls_m1, cls_ones


@pytest.fixture(scope="module")
def programs():
    """(K, ARITH) -> Program, from the preprocessed source: the statements with operands w0..w<K-1>; ARITH ones build the
    class mask with s_andn2_b32 / s_lshl_b64 instead of reading it from the lanes."""
    progs = {}
    for text, operands in preprocessed_asm_templates():
        if "v_readlane_b32 s44, %[ex0], m0" not in text:
            continue
        K = 1 + max(int(c) for c in "0123" if "[w%s]" % c in operands)
        arith = "s_andn2_b32 s48, s56, %[sm1]" in text or (K == 1 and (K, True) not in progs)
        progs[(K, arith)] = Program(text)
    assert sorted(progs) == [(k, a) for k in (1, 2, 3, 4) for a in (False, True)], sorted(progs)
    return progs


@pytest.mark.parametrize("K", [1, 2, 3, 4])
@pytest.mark.parametrize("arith", [True, False])
def test_scalar_loop_equals_the_automaton(programs, K, arith):
    rng = random.Random(1000 * K + arith)
    checked = clean = total_exits = total_steps = total_ins = 0
    for _ in range(300):
        P, ex, cls_mask, nb, cls_m1, cls_ones = random_case(rng, arith)
        want = reference(K, P, ex, cls_mask, nb)
        got = with_asm(programs[(K, arith)], K, P, ex, cls_mask, nb, cls_m1, cls_ones)
        assert got[3] == want[3]                                 # the launch fails in the same cases
        if want[3]:
            continue
        checked += 1
        for c in range(K):
            assert got[0][c][:nb] == want[0][c][:nb], (K, arith, c)
        # (the assembly does not drop an emptied lowest plane itself: when the batch's last step empties it, that is left
        # to the first step of the next batch -- one pending drop is the same state)
        planes, shifts = [list(p) for p in got[1]], got[2]
        shifts += drop_empty_plane(planes)
        assert (planes, shifts) == (want[1], want[2]) or (got[1], got[2]) == (want[1], want[2])
        clean += got[4] == 0
        total_exits += got[4]
        total_steps += nb
        total_ins += got[5]
    assert checked > 100
    assert clean > 10                                            # whole batches inside the assembly, too
    # the common path is what the design says it is: about 15 + 13 K instructions per step
    assert total_ins / total_steps < 20 + 16 * K


def test_word0_path_length_of_the_headline_instantiation(programs):
    """K = 2 with arithmetic class masks (config 3): a step whose two picks come from word 0 of plane 0 executes 26
    instructions (4 lane reads, 3 + 4 + 3 + 1 for the first pick with its class mask, 3 + 4 + 1 for the second, 3 of the
    loop) -- the figure DESIGN.md section 4.1a and the cost model of section 10 are built on."""
    S = 16
    cls_mask = [((1 << S) - 1) << (l & ~(S - 1)) if l < 64 else 0 for l in range(128)]
    P = [[(1 << 64) - 1, 0], [0, 0], [0, 0], [0, 0]]
    ex = [cls_mask[(5 * r) % 64] for r in range(64)]
    got = with_asm(programs[(2, True)], 2, P, ex, cls_mask, 16, S - 1, (1 << S) - 1)
    assert got[4] == 0 and not got[3]
    assert got[5] == 1 + 26 * 16 + 1 + 3, got[5]                 # s_mov m0 | 16 steps | s_branch 7f | the three moves at 7:


def _dpp_programs():
    wave = rows4 = rows16 = None
    for text, operands in preprocessed_asm_templates():
        if "v_min_u32_dpp" not in text:
            continue
        if "row_bcast:31" in text:
            wave = wave or Program(text)
        elif "row_mirror" in text:
            rows16 = rows16 or Program(text)                     # the second half of row_min_u32<16>
        elif "quad_perm" in text:
            rows4 = rows4 or Program(text)
    assert wave and rows4 and rows16
    return wave, rows4, rows16


def test_dpp_minima_of_dev_common():
    """wave_min_u32_bcast and row_min_u32 (dev_common.h) are inline assembly on the device and builtins in the emulator: the
    assembly text, executed under the documented DPP semantics, leaves the wave's minimum in lane 63 and each group's
    minimum in all of its lanes."""
    wave, rows4, rows16 = _dpp_programs()
    rng = random.Random(7)
    for it in range(200):
        v = [rng.choice([rng.getrandbits(32), rng.getrandbits(8), 0xffffffff]) for _ in range(64)]
        ops = {"0": list(v)}
        Machine(ops).run(wave)
        assert ops["0"][63] == min(v)
        ops = {"0": list(v)}
        Machine(ops).run(rows4)
        assert ops["0"] == [min(v[l & ~3:(l & ~3) + 4]) for l in range(64)]
        Machine(ops).run(rows16)                                 # row_min_u32<16> = the quad part, then the row part
        assert ops["0"] == [min(v[l & ~15:(l & ~15) + 16]) for l in range(64)]
