"""blance_upload / blance_download (ABI 5): arrays in page-locked memory of blance_host_alloc go by DMA where they lie,
pageable ones through the context's staging buffer; the O(P) part of blance_validate runs on the device.  Same results
and the same refusals either way.  CPU: the product's host code over the SIMT emulator; `-m gpu`: the device."""
import ctypes as C

import numpy as np
import pytest

from blance_amd import abi, hip, synth
from test_simt_emulated import build_emu


@pytest.fixture(scope="module")
def emu_lib():
    return build_emu()


def _oracle(fp):
    from oracle import loader
    return loader.plan(fp)


def _check_both_ways(lib_path, fp, **kw):
    want = _oracle(fp).digest()
    pl = hip.Planner(lib_path=lib_path, **kw)
    assert pl.plan(fp).digest() == want                      # pageable in, pageable out (staged)
    arena = hip.HostArena(lib_path)
    pl.upload(fp.pin(arena))                                 # page-locked in ...
    pl.plan_resident()
    res = pl.download(arena)                                 # ... and out
    assert res.digest() == want
    assert pl.download().digest() == want                    # page-locked in, pageable out
    pl.close()
    arena.close()


def test_pinned_and_staged_agree_emulated(emu_lib):
    _check_both_ways(emu_lib, synth.config_flat(3, P=1024, N=128), chain_min_parts=8)
    fp1 = synth.config5_initial(1500, 96)
    _check_both_ways(emu_lib, fp1)


def test_host_alloc_blocks_are_reused(emu_lib):
    lib = hip.load_library(emu_lib)
    p1 = lib.blance_host_alloc(1 << 20)
    assert p1
    lib.blance_host_free(p1)
    p2 = lib.blance_host_alloc((1 << 20) - 4096)             # fits the cached block
    assert p2 == p1
    lib.blance_host_free(p2)
    lib.blance_host_free(None)
    lib.blance_host_free(C.c_void_p(12345))                  # not one of the library's: ignored


def test_host_trim_and_last_context_release_the_cache(emu_lib):
    """(ABI 6) blance_host_trim() returns the cached page-locked blocks to the system; so does the destruction of the
    process's last context.  Blocks the caller still holds stay valid."""
    lib = hip.load_library(emu_lib)
    held = lib.blance_host_alloc(1 << 20)
    p1 = lib.blance_host_alloc(3 << 20)
    lib.blance_host_free(p1)
    lib.blance_host_trim()
    (C.c_char * 16).from_address(held)[:4] = b"abcd"         # still the caller's
    lib.blance_host_free(held)
    pl = hip.Planner(lib_path=emu_lib)
    fp = synth.config_flat(2, P=300, N=20)
    assert pl.plan(fp).digest() == _oracle(fp).digest()
    pl.close()                                               # the last context of this process (if it is): cache dropped
    p3 = lib.blance_host_alloc(1 << 20)
    assert p3
    lib.blance_host_free(p3)


def test_arena_arrays_own_their_blocks(emu_lib):
    """An array of a HostArena keeps its block: a result kept after the arena (and the FlatResult) are gone never aliases a
    block that was handed out again; FlatProblem.pin() returns a copy and leaves the problem it was called on alone."""
    import gc
    fp = synth.config_flat(3, P=512, N=128)
    before = {n: a.ctypes.data for n, a in fp.arrays.items()}
    arena = hip.HostArena(emu_lib)
    fpp = fp.pin(arena)
    assert {n: a.ctypes.data for n, a in fp.arrays.items()} == before and fpp is not fp
    assert all(np.array_equal(fpp.arrays[n], fp.arrays[n]) for n in fp.arrays)
    pl = hip.Planner(lib_path=emu_lib, chain_min_parts=8)
    pl.upload(fpp)
    pl.plan_resident()
    res = pl.download(arena)
    want = _oracle(fp)
    nodes, snapshot = res.out_nodes, res.out_nodes.copy()
    del res, fpp
    arena.close()
    del arena
    gc.collect()
    other = hip.HostArena(emu_lib)                            # new allocations of the same sizes: must not land on `nodes`
    junk = [other.empty(nodes.size, np.int32) for _ in range(4)]
    for j in junk:
        j[...] = -7
    assert np.array_equal(nodes, snapshot) and np.array_equal(nodes[:want.out_nodes.size], want.out_nodes)
    pl.close()


BAD = [("long then not monotone", lambda fp: fp.arrays["assign_off"].__setitem__(3, 10 ** 6), abi.ERR_UNSUPPORTED),
       ("not monotone", lambda fp: fp.arrays["prev_off"].__setitem__(7, -5), abi.ERR_BAD_ARG),
       ("kind", lambda fp: fp.arrays["prev_kind"].__setitem__(5, 7), abi.ERR_BAD_ARG),
       ("order", lambda fp: fp.arrays["part_order"].__setitem__(0, int(fp.arrays["part_order"][1])), abi.ERR_BAD_ARG),
       ("order range", lambda fp: fp.arrays["part_order"].__setitem__(2, -1), abi.ERR_BAD_ARG),
       ("prev id", lambda fp: fp.arrays["prev_nodes"].__setitem__(0, 10 ** 6), abi.ERR_BAD_ARG),
       ("assign id", lambda fp: fp.arrays["assign_nodes"].__setitem__(1, -3), abi.ERR_BAD_ARG),
       ("weights", lambda fp: fp.arrays["part_weight"].__setitem__(slice(None), 2 ** 30), abi.ERR_UNSUPPORTED)]


def _rebalance_problem():
    from oracle import loader
    fp1 = synth.config5_initial(600, 48)
    return synth.config5_rebalance(fp1, loader.plan(fp1), 600, 48)


@pytest.mark.parametrize("name,spoil,status", BAD, ids=[b[0] for b in BAD])
def test_device_side_validation_refuses_what_the_host_refuses(emu_lib, name, spoil, status):
    fp = _rebalance_problem()
    for a in fp.arrays.values():
        a.setflags(write=True)
    spoil(fp)
    fp._struct = None
    lib = hip.load_library(emu_lib)
    assert lib.blance_validate(C.byref(fp.as_struct())) == status
    host_text = lib.blance_last_error()
    pl = hip.Planner(lib_path=emu_lib)
    with pytest.raises(hip.BlanceError) as e:
        pl.upload(fp)
    assert e.value.status == status
    assert host_text.decode() in str(e.value)
    # the context is still usable
    good = _rebalance_problem()
    assert pl.plan(good).digest() == _oracle(good).digest()
    pl.close()


@pytest.mark.gpu
def test_pinned_and_staged_agree_on_the_device():
    _check_both_ways(None, synth.config_flat(3, P=65536, N=1024))
    _check_both_ways(None, synth.config_flat(2, P=65536, N=256))
    fp1 = synth.config5_initial(20000, 512)
    _check_both_ways(None, fp1)


@pytest.mark.gpu
@pytest.mark.parametrize("name,spoil,status", BAD, ids=[b[0] for b in BAD])
def test_device_side_validation_on_the_device(name, spoil, status):
    fp = _rebalance_problem()
    for a in fp.arrays.values():
        a.setflags(write=True)
    spoil(fp)
    fp._struct = None
    pl = hip.Planner()
    with pytest.raises(hip.BlanceError) as e:
        pl.upload(fp)
    assert e.value.status == status
    good = _rebalance_problem()
    assert pl.plan(good).digest() == _oracle(good).digest()
    pl.close()
