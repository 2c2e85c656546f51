"""Bounds, alignment and bijectivity of the kernels' index arithmetic
(csrc/common/index_math.h), checked on the host over the whole domain the kernels
use.  A wrong LDS swizzle or tile order would not fail loudly on the GPU, so the
maps are compiled for the host as well and tested exhaustively here."""

import ctypes

import numpy as np
import pytest

from nanofed_amd import native_build


@pytest.fixture(scope="module")
def ix():
    native_build.build_host()
    lib = ctypes.CDLL(str(native_build.HOST_LIB))
    for name in ("gemm_kc_off", "gemm_km_off", "gemm_tile_index", "gemm_tile_rc", "mnist_fwd_off", "mnist_fwd_block_dy", "mnist_fwd_block_dx",
                 "mnist_dg_off", "lds_b128_group",
                 "mfma32_acc_row", "conv_patch_source", "conv_patch_index", "conv_patch_byte",
                 "conv_patch_frag_byte", "conv_patch_read_cycles", "conv_patch_row_pitch",
                 "conv_patch_band_pitch"):
        getattr(lib, f"nf_ix_{name}").restype = ctypes.c_int
    return lib


@pytest.mark.parametrize("rows", [128, 256])
def test_gemm_k_contiguous_tile_is_a_bijection_onto_its_bytes(ix, rows):
    # a stage holds `rows` rows of 64 k = 8 pieces of 16 bytes
    offs = np.array([[ix.nf_ix_gemm_kc_off(r, c) for c in range(8)] for r in range(rows)])
    assert offs.min() == 0 and offs.max() == rows * 128 - 16 and (offs % 16 == 0).all()
    assert len(np.unique(offs)) == rows * 8
    # a piece never leaves its row (the DMA writes rows linearly, the swizzle permutes within)
    assert ((offs // 128) == np.arange(rows)[:, None]).all()
    # the 8 lanes that read one k piece of 8 consecutive row pairs hit 8 distinct bank groups
    for chunk in range(8):
        groups = {(ix.nf_ix_gemm_kc_off(r, chunk) % 128) // 16 for r in range(0, 16, 2)}
        assert len(groups) == 8


@pytest.mark.parametrize("x", [128, 256])
def test_gemm_k_major_tile_is_a_bijection_onto_its_bytes(ix, x):
    rowb = x * 2  # bf16
    offs = np.array([[ix.nf_ix_gemm_km_off(rowb, k, xb) for xb in range(0, rowb, 8)] for k in range(64)])
    assert offs.min() == 0 and offs.max() == 64 * rowb - 8 and (offs % 8 == 0).all()
    assert len(np.unique(offs)) == offs.size
    assert ((offs // rowb) == np.arange(64)[:, None]).all()
    # 16-byte pieces stay whole: byte xb and xb + 8 of an aligned piece are neighbours
    for k in range(64):
        for xb in range(0, rowb, 16):
            assert ix.nf_ix_gemm_km_off(rowb, k, xb + 8) == ix.nf_ix_gemm_km_off(rowb, k, xb) + 8


@pytest.mark.parametrize("total", [1, 7, 8, 9, 36, 63, 64, 65, 200, 393, 1571])
def test_every_tile_is_owned_by_exactly_one_workgroup(ix, total):
    tiles = [ix.nf_ix_gemm_tile_index(i, total) for i in range(total)]
    assert sorted(tiles) == list(range(total))
    # the ids of one XCD (congruent mod 8) own one contiguous range
    for xcd in range(min(8, total)):
        mine = [tiles[i] for i in range(xcd, total, 8)]
        assert mine == list(range(mine[0], mine[0] + len(mine)))


@pytest.mark.parametrize("tiles_m,tiles_n", [(16, 12), (16, 197), (3, 5), (1, 9), (7, 1), (10, 3)])
@pytest.mark.parametrize("group_m", [1, 2, 4, 8, 16])
def test_grouped_tile_order_visits_every_tile_once(ix, tiles_m, tiles_n, group_m):
    seen = [ix.nf_ix_gemm_tile_rc(t, tiles_m, tiles_n, group_m) for t in range(tiles_m * tiles_n)]
    cells = [(rc >> 16, rc & 65535) for rc in seen]
    assert sorted(cells) == [(i, j) for i in range(tiles_m) for j in range(tiles_n)]
    if group_m > 1 and tiles_m >= group_m:
        # the first group_m tiles are one column of the first band
        assert cells[:group_m] == [(i, 0) for i in range(group_m)]


def _slots_distinct(addresses_by_lane, group_of):
    """Every ds_read_b128 lane group (16 lanes) touches 16 different 16-byte slots of the
    256-byte bank row."""
    groups = {}
    for lane, addr in addresses_by_lane.items():
        groups.setdefault(group_of(lane), []).append((addr % 256) // 16)
    return all(len(s) == 16 and len(set(s)) == 16 for s in groups.values())


def test_mnist_forward_band_image(ix):
    # a band: 10 rows of 26 pixels, 64 bytes each, rows 1680 bytes apart
    offs = np.array([[[ix.nf_ix_mnist_fwd_off(r, c, ch) for ch in range(4)] for c in range(26)]
                     for r in range(10)])
    assert offs.min() == 0 and offs.max() == 9 * 1680 + 25 * 64 + 48 and (offs % 16 == 0).all()
    assert len(np.unique(offs)) == offs.size and offs.max() + 16 <= 64 * 592
    # the staging pieces (104 per row) and the fragment reads address the same bytes
    for piece in range(1040):
        row, col16 = divmod(piece, 104)
        assert row * 1680 + col16 * 16 == ix.nf_ix_mnist_fwd_off(row, col16 // 4, col16 % 4)
    # rows of a block: every pooling window (4 consecutive rows) is a 2 x 2 pixel square
    pos = [(ix.nf_ix_mnist_fwd_block_dy(m), ix.nf_ix_mnist_fwd_block_dx(m)) for m in range(32)]
    assert sorted(pos) == [(y, x) for y in range(4) for x in range(8)]
    for w in range(8):
        ys = {pos[4 * w + q][0] for q in range(4)}
        xs = {pos[4 * w + q][1] for q in range(4)}
        assert len(ys) == 2 and len(xs) == 2 and max(ys) - min(ys) == 1 and max(xs) - min(xs) == 1
    # conflict-free fragment reads for every block position, tap and k half
    for mgrp in range(2):
        for i in range(3):
            for kh in range(3):
                for kw in range(3):
                    for half in range(2):
                        addr = {
                            lane: ix.nf_ix_mnist_fwd_off(4 * mgrp + pos[lane & 31][0] + kh,
                                                        8 * i + pos[lane & 31][1] + kw,
                                                        2 * half + (lane >> 5))
                            for lane in range(64)
                        }
                        assert max(addr.values()) + 16 <= 10 * 1680
                        assert _slots_distinct(addr, ix.nf_ix_lds_b128_group)


def test_mnist_dgrad_patch(ix):
    offs = np.array([[[ix.nf_ix_mnist_dg_off(pr, pc, part) for part in range(8)]
                      for pc in range(28)] for pr in range(13)])
    assert offs.min() == 0 and (offs % 16 == 0).all() and offs.max() + 16 <= 13 * 4256
    assert len(np.unique(offs)) == offs.size
    # blocks of 32 consecutive pixels of the 26-wide image, shifted by any tap: one LDS
    # cycle per lane group, across the ends of image rows as well
    for mb in range(22):
        for kh in range(3):
            for kw in range(3):
                for ks in range(4):
                    addr = {}
                    for lane in range(64):
                        p = min(mb * 32 + (lane & 31), 675)
                        y, x = divmod(p, 26)
                        addr[lane] = ix.nf_ix_mnist_dg_off(y + 2 - kh, x + 2 - kw,
                                                           2 * ks + (lane >> 5))
                    if mb * 32 + 31 <= 675:  # (the clamped tail block repeats one pixel)
                        assert _slots_distinct(addr, ix.nf_ix_lds_b128_group)


def test_accumulator_rows_cover_the_tile_once(ix):
    seen = set()
    for lane in range(64):
        for reg in range(16):
            row = ix.nf_ix_mfma32_acc_row(reg, lane)
            assert 0 <= row < 32
            seen.add((row, lane & 31))
    assert len(seen) == 32 * 32


@pytest.mark.parametrize("H,W,RI,NI", [(32, 32, 4, 1), (16, 16, 8, 1), (8, 8, 8, 2), (4, 4, 4, 8)])
def test_conv_patch_maps_stay_inside_the_tensor_and_the_patch(ix, H, W, RI, NI):
    B, Cin = 16, 64  # a tile spans NI * RI image rows; the batch is a multiple of that
    logH, logW, logRI = H.bit_length() - 1, W.bit_length() - 1, RI.bit_length() - 1
    patch = NI * (RI + 2) * (W + 2)
    rows_per_tile = NI * RI
    for row0 in range(0, B * H, rows_per_tile):
        sources = [ix.nf_ix_conv_patch_source(H, W, logH, RI, Cin, row0, pp) for pp in range(patch)]
        inside = [s for s in sources if s >= 0]
        assert all(0 <= s <= (B * H * W - 1) * Cin and s % Cin == 0 for s in inside)
        assert len(set(inside)) == len(inside)
        # every output pixel's 3x3 neighbourhood lies inside the patch and its centre is the pixel
        for m in range(rows_per_tile * W):
            p = ix.nf_ix_conv_patch_index(W, logW, RI, logRI, m)
            corners = [p + dy * (W + 2) + dx for dy in range(3) for dx in range(3)]
            assert 0 <= min(corners) and max(corners) < patch
            centre = sources[p + (W + 2) + 1]
            grow = row0 + m // W
            assert centre == ((grow // H * H + grow % H) * W + m % W) * Cin


@pytest.mark.parametrize("pitch,tp", [(80, 128), (144, 64)])
@pytest.mark.parametrize("H,W", [(32, 32), (16, 16), (8, 8), (4, 4)])
def test_padded_conv_patch_is_disjoint_and_its_fragment_reads_conflict_free(ix, H, W, pitch, tp):
    rows = tp // W
    RI = min(rows, H)
    NI = rows // RI
    logW, logRI = W.bit_length() - 1, RI.bit_length() - 1
    args = (W, logW, RI, logRI, NI, tp, pitch)
    rp, bp = ix.nf_ix_conv_patch_row_pitch(*args), ix.nf_ix_conv_patch_band_pitch(*args)
    assert rp % 16 == 0 and bp % 16 == 0
    assert (W + 2) * pitch <= rp < (W + 2) * pitch + 256 and (RI + 2) * rp <= bp < (RI + 2) * rp + 256
    patch = NI * (RI + 2) * (W + 2)
    # every patch pixel owns pitch - 16 bytes of its own inside the image
    starts = sorted(ix.nf_ix_conv_patch_byte(W, RI, pp, pitch, rp, bp) for pp in range(patch))
    assert starts[0] == 0 and starts[-1] + pitch - 16 <= NI * bp
    assert all(b - a >= pitch for a, b in zip(starts, starts[1:]))
    assert all(s % 16 == 0 for s in starts)
    # an output pixel's taps are the staged pixels around it
    staged = {ix.nf_ix_conv_patch_byte(W, RI, pp, pitch, rp, bp): pp for pp in range(patch)}
    for m in range(tp):
        base = ix.nf_ix_conv_patch_frag_byte(W, logW, RI, logRI, m, pitch, rp, bp)
        p = ix.nf_ix_conv_patch_index(W, logW, RI, logRI, m)
        for ky in range(3):
            for kx in range(3):
                assert staged[base + ky * rp + kx * pitch] == p + ky * (W + 2) + kx
    # one LDS cycle per lane group: 2 per block of 32 pixels; the plain layout needs more
    # for images narrower than 32 pixels
    assert ix.nf_ix_conv_patch_read_cycles(W, logW, RI, logRI, tp, pitch, rp, bp) == 2 * (tp // 32)
    plain_rp = (W + 2) * pitch
    plain = ix.nf_ix_conv_patch_read_cycles(W, logW, RI, logRI, tp, pitch, plain_rp,
                                            (RI + 2) * plain_rp)
    assert plain == 2 * (tp // 32) if W == 32 else plain > 2 * (tp // 32)
