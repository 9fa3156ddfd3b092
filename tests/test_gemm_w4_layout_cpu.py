"""Index logic of csrc/gemm16_w4.hip restated lane by lane in numpy (no GPU):

  * the LDS image of a K-tile as the DMA pieces lay it down (8 rows x 128 B per piece, lane -> (row, physical chunk), source-side XOR
    swizzle) and the fragment reads that take it apart again (row tile i, k-step kk, lane -> 16 bytes): every lane must receive the eight
    k-values of ITS fragment row, and the 16 lanes the hardware services together (MI355X_MICROARCH.md, LDS table: ds_read_b128 lane groups)
    must fall on 16 different 16-byte bank slots;
  * the MFMA operand order (first source = W fragment, second = X fragment) and the accumulator -> (row, column) map of the epilogue through
    the slab: written in the accumulator layout, read as row lines; against C = X W^T on a whole 256 x 256 x 128 tile;
  * the slab reads (ds_read_b128) are conflict-free in their lane groups; the slab writes (ds_write_b64: 4 x 16 contiguous lanes over 32 banks)
    are exactly 2-way -- rows r and r + 8 of a step share their banks -- which the test pins as the known cost of this layout;
  * the tile dealing (XCD-contiguous ranges, left-over tiles to the first workgroups) covers every tile exactly once for any grid."""
import numpy as np
import pytest

BUFB, BOFF = 65536, 32768
B128_GROUPS = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)), list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32))]
B128_GROUPS += [[l + 32 for l in g] for g in B128_GROUPS]


def stage_ktile(lds, buf, X, W, m0, n0, kt):
    """What the four waves' 16 DMA pieces of one K-tile write: element (not byte) granular emulation, 2 bytes per element."""
    for wave in range(4):
        for p in range(8):
            for op, src, r0, base in ((0, X, m0, 0), (1, W, n0, BOFF)):
                dst = buf * BUFB + base + (wave * 64 + p * 8) * 128                 # wave-uniform LDS byte address of the piece
                for lane in range(64):
                    lrow, pc = lane >> 3, lane & 7
                    row = r0 + wave * 64 + p * 8 + lrow
                    lc = pc ^ lrow                                                  # logical chunk fetched by this lane
                    k0 = kt * 64 + lc * 8
                    a = dst + lane * 16                                             # lane-linear landing
                    lds[a // 2:a // 2 + 8] = src[row, k0:k0 + 8]


def frag_addr(op, wave, tile, kk, lane, buf):
    """Byte address read by `lane` for fragment `tile` (row tile of X for op 0, column tile of W for op 1) of k-step kk."""
    wr, wc = wave >> 1, wave & 1
    frow, fq, fsw = lane & 15, lane >> 4, lane & 7
    off = ((kk * 4 + fq) ^ fsw) * 16
    if op == 0:
        return buf * BUFB + (wr * 128 + frow) * 128 + off + tile * 2048
    return buf * BUFB + BOFF + (wc * 128 + frow) * 128 + off + tile * 2048


def test_fragment_reads_return_the_right_rows_and_are_conflict_free():
    rng = np.random.default_rng(0)
    X = rng.integers(0, 1 << 15, size=(512, 192)).astype(np.float64)            # tagged values: exact in float
    W = rng.integers(0, 1 << 15, size=(512, 192)).astype(np.float64)
    lds = np.full(2 * BUFB // 2, -1.0)
    m0, n0, kt = 256, 0, 2
    stage_ktile(lds, 1, X, W, m0, n0, kt)
    assert (lds[BUFB // 2:] >= 0).all()                                          # the 16 x 4 pieces fill the whole 64 KB buffer
    for wave in range(4):
        wr, wc = wave >> 1, wave & 1
        for op, src, r0 in ((0, X, m0 + wr * 128), (1, W, n0 + wc * 128)):
            for tile in range(8):
                for kk in range(2):
                    addrs = [frag_addr(op, wave, tile, kk, lane, 1) for lane in range(64)]
                    for lane, a in enumerate(addrs):
                        assert a % 16 == 0
                        row, k0 = r0 + tile * 16 + (lane & 15), kt * 64 + kk * 32 + (lane >> 4) * 8
                        assert (lds[a // 2:a // 2 + 8] == src[row, k0:k0 + 8]).all(), (op, wave, tile, kk, lane)
                    for grp in B128_GROUPS:                                      # 64 banks x 4 B: 16 lanes x 16 B must tile the 256 B
                        slots = {(addrs[l] // 16) % 16 for l in grp}
                        assert len(slots) == 16, (op, wave, tile, kk, sorted(slots))


def _mfma(first, second, acc):
    """v_mfma_f32_16x16x32: D[i][j] += sum_k first[i][k] second[j][k]; lane holds row (lane & 15), k = (lane >> 4) * 8 + [0, 8) of either
    operand and D rows (lane >> 4) * 4 + [0, 4) at column lane & 15."""
    A, Bm = np.zeros((16, 32)), np.zeros((16, 32))
    for lane in range(64):
        A[lane & 15, (lane >> 4) * 8:(lane >> 4) * 8 + 8] = first[lane]
        Bm[lane & 15, (lane >> 4) * 8:(lane >> 4) * 8 + 8] = second[lane]
    D = A @ Bm.T
    out = acc.copy()
    for lane in range(64):
        out[lane] += D[(lane >> 4) * 4:(lane >> 4) * 4 + 4, lane & 15]
    return out


def test_tile_product_and_epilogue_map_against_the_closed_form():
    rng = np.random.default_rng(1)
    K = 128
    X = rng.integers(-3, 4, size=(256, K)).astype(np.float64)
    W = rng.integers(-3, 4, size=(256, K)).astype(np.float64)
    ref = X @ W.T
    lds = np.zeros(2 * BUFB // 2)
    for kt in range(K // 64):
        stage_ktile(lds, kt & 1, X, W, 0, 0, kt)
    out = np.full((256, 256), np.nan)
    for wave in range(4):
        wr, wc = wave >> 1, wave & 1
        acc = np.zeros((64, 64, 4))                                              # [m = i * 8 + j][lane][r]
        for kt in range(K // 64):
            for kk in range(2):
                fa = [[lds[frag_addr(0, wave, i, kk, lane, kt & 1) // 2:][:8] for lane in range(64)] for i in range(8)]
                fb = [[lds[frag_addr(1, wave, j, kk, lane, kt & 1) // 2:][:8] for lane in range(64)] for j in range(8)]
                for i in range(8):
                    for j in range(8):
                        acc[i * 8 + j] = _mfma(np.array(fb[j]), np.array(fa[i]), acc[i * 8 + j])     # first source = W fragment
        # epilogue: 16 steps (jh, i); slab = 16 rows x 128 B of 16-bit values, element granular here
        for s in range(16):
            jh, i = s >> 3, s & 7
            slab = np.full(16 * 64, np.nan)
            waddr = []
            for lane in range(64):
                l15, fq4 = lane & 15, lane >> 4
                for jj in range(4):
                    a = l15 * 128 + (((jj * 2 + (fq4 >> 1)) ^ (l15 & 7)) * 16) + (fq4 & 1) * 8
                    waddr.append((jj, lane, a))
                    slab[a // 2:a // 2 + 4] = acc[i * 8 + jh * 4 + jj][lane]
            assert not np.isnan(slab).any()
            for jj in range(4):                                                  # ds_write_b64: 4 groups of 16 contiguous lanes, 32 banks x 4 B
                for g in range(4):                                               # rows r and r + 8 share their banks: exactly 2-way (the layout of
                    hits = {}                                                    # gemm16_p8's slab; 2 extra LDS cycles per write, 64 writes per tile)
                    for lane in range(16 * g, 16 * g + 16):
                        a = [w for w in waddr if w[0] == jj and w[1] == lane][0][2]
                        for bnk in ((a // 4) % 32, (a // 4 + 1) % 32):
                            hits[bnk] = hits.get(bnk, 0) + 1
                    assert len(hits) == 16 and set(hits.values()) == {2}, (jj, g, hits)
            for h in range(2):
                raddr = []
                for lane in range(64):
                    srow, sch = lane >> 3, lane & 7
                    r = h * 8 + srow
                    a = r * 128 + ((sch ^ (r & 7)) * 16)
                    raddr.append(a)
                    m, n = wr * 128 + i * 16 + r, wc * 128 + jh * 64 + sch * 8
                    out[m, n:n + 8] = slab[a // 2:a // 2 + 8]
                for grp in B128_GROUPS:
                    assert len({(raddr[l] // 16) % 16 for l in grp}) == 16
    assert (out == ref).all()


@pytest.mark.parametrize("ntiles,ncu", [(1773, 256), (72, 256), (270, 256), (256, 256), (2364, 256), (591, 304), (100, 7), (33, 32)])
def test_tile_dealing_covers_every_tile_once(ntiles, ncu):
    grid = min(ntiles, ncu)
    full, left = ntiles // grid, ntiles - (ntiles // grid) * grid
    seen = np.zeros(ntiles, dtype=int)
    for blk in range(grid):
        xcd, slot = blk & 7, blk >> 3
        gq, gr = grid >> 3, grid & 7
        per_xcd = gq + (1 if xcd < gr else 0)
        my_first = full * (xcd * gq + min(xcd, gr)) + slot
        count = full + (1 if blk < left else 0)
        for e in range(count):
            tile = my_first + e * per_xcd if e < full else full * grid + blk
            seen[tile] += 1
        # an XCD's workgroups work on neighbouring tiles at every step e: its tiles of step e are consecutive
    assert (seen == 1).all()
    for xcd in range(min(8, grid)):
        for e in range(full):
            tiles = sorted(full * (xcd * (grid >> 3) + min(xcd, grid & 7)) + slot + e * ((grid >> 3) + (1 if xcd < (grid & 7) else 0))
                           for slot in range((grid >> 3) + (1 if xcd < (grid & 7) else 0)))
            assert tiles == list(range(tiles[0], tiles[0] + len(tiles)))
