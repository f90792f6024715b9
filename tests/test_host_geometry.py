"""CPU suite: launch geometry of the blocked sweep (xpoly_amd/csrc/lp_blocked.hip.h blk_sweep_tile / blk_sweep_grid,
lp_host.hip.h pick_ld) through the library's host-side views -- no device needed. The GPU tests can only cover the
shapes they run; the map must place every tile of ANY strips x rowblocks tableau on exactly one workgroup."""
import ctypes as C

import pytest

from xpoly_amd._capi import lib


def tiles(strips, rowblocks, rev):
    L = lib()
    bx, by = C.c_int(), C.c_int()
    grid = L.xpg_test_sweep_tile(strips, rowblocks, rev, -1, C.byref(bx), C.byref(by))
    out = []
    for lid in range(grid):
        live = L.xpg_test_sweep_tile(strips, rowblocks, rev, lid, C.byref(bx), C.byref(by))
        assert live in (0, 1)
        if live:
            out.append((lid, bx.value, by.value))
    return grid, out


@pytest.mark.parametrize("strips", list(range(1, 36)) + [40, 63, 64, 65])
def test_sweep_tile_map_covers_every_tile_once(strips):
    for rowblocks in (1, 2, 7, 8, 9, 16, 63, 64, 256):
        for rev in (0, 1):
            grid, t = tiles(strips, rowblocks, rev)
            assert grid >= strips * rowblocks and grid % 8 == 0
            seen = {(x, y) for _, x, y in t}
            assert len(t) == len(seen) == strips * rowblocks, (strips, rowblocks, rev)
            assert seen == {(x, y) for x in range(strips) for y in range(rowblocks)}
            # the point of the map: a strip below 8 * floor(strips / 8) always lands on workgroup ids of ONE residue mod 8
            full = 8 * (strips // 8)
            for lid, x, y in t:
                if x < full:
                    assert lid % 8 == x % 8, (strips, rowblocks, lid, x)
            # and the leftover strips are spread evenly: every residue gets rowblocks / 8 of each, give or take one
            for q in range(full, strips):
                per = [0] * 8
                for lid, x, y in t:
                    if x == q:
                        per[lid % 8] += 1
                assert max(per) - min(per) <= 1, (strips, rowblocks, q, per)


def test_sweep_direction_reverses_row_blocks_only():
    for strips, rowblocks in ((16, 256), (25, 256), (3, 10)):
        _, fwd = tiles(strips, rowblocks, 0)
        _, bwd = tiles(strips, rowblocks, 1)
        assert [(l, x, rowblocks - 1 - y) for l, x, y in fwd] == bwd


def test_leading_dimension_rule():
    L = lib()
    for W in list(range(1, 300)) + [4095, 4096, 4097, 4112, 8191, 8192, 8193, 8224, 12288, 12289, 12336, 16385, 16448, 32769]:
        ld = L.xpg_test_pick_ld(W)
        assert ld >= W and ld % 16 == 0 and ld - W < 96, (W, ld)
        assert ld % 4112 != 0 and (ld + 16) % 4096 != 0, (W, ld)        # the two row strides the sweep runs slowly at
        if W % 16 == 0 and W % 4112 != 0 and (W + 16) % 4096 != 0:
            assert ld == W                                               # 4096 x 8192 keeps ld = 8192
    assert L.xpg_test_pick_ld(8192) == 8192 and L.xpg_test_pick_ld(12289) == 12352
    for W in range(1, 20000, 7):                                          # build() relies on this (phase-1 column: W and W + 1)
        assert L.xpg_test_pick_ld(W) <= max(L.xpg_test_pick_ld(W + 1), L.xpg_test_pick_ld(W))
