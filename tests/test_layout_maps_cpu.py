"""CPU: the LDS images the kernels rely on, restated as index arithmetic and checked against the bank model of
MI355X_MICROARCH.md (64 banks x 4 B; ds_read_b128 is served in four fixed 16-lane groups, ds_read_b64_tr_b16 in two 32-lane groups;
lanes of one group conflict when they touch the same bank at different addresses).  Pins the "conflict-free" claims of
csrc/attention32.hip (rot4 chunk swizzle), csrc/gemm.hip / csrc/gemm_pp.hip (chunk ^ row&7; staging units and epilogue lane pairing of the
ping-pong GEMM), and the row <-> fragment bijections."""
import itertools

B128_GROUPS = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)),
               list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32)),
               list(range(32, 36)) + list(range(44, 48)) + list(range(52, 60)),
               list(range(36, 44)) + list(range(48, 52)) + list(range(60, 64))]
TR_GROUPS = [list(range(0, 32)), list(range(32, 64))]


def conflicts(addr_of_lane, groups, nbytes):
    """number of (group, bank) pairs hit by two different dword addresses"""
    bad = 0
    for g in groups:
        seen = {}
        for lane in g:
            a = addr_of_lane(lane)
            assert a % min(nbytes, 16) == 0
            for dw in range(nbytes // 4):
                bank, word = ((a // 4) + dw) % 64, (a // 4) + dw
                if seen.setdefault(bank, word) != word:
                    bad += 1
    return bad


def rot4(r):
    return ((r & 3) << 2) | ((r >> 2) & 3)


def test_attention32_tile_swizzle_is_conflict_free():
    """tile [64 rows][256 B], 16-byte chunk c of row r at chunk c ^ rot4(r) (attention32.hip LaneAddr)"""
    KP = 256
    for ks, blk in itertools.product(range(8), range(2)):          # row fragments: lane -> row l&31 of a 32-row block, chunk 2 ks + hi
        def rm(lane, ks=ks, blk=blk):
            l31, hi = lane & 31, lane >> 5
            return (blk * 32 + l31) * KP + (((ks * 2 + hi) ^ rot4(l31)) << 4)
        assert conflicts(rm, B128_GROUPS, 16) == 0
    for db, half, grp in itertools.product(range(4), range(2), range(4)):      # transpose reads of a 16-row group
        def tr(lane, db=db, half=half, grp=grp):
            l31, hi, i16 = lane & 31, lane >> 5, lane & 15
            r = half * 8 + 4 * hi + (i16 >> 2)
            chunk = db * 4 + ((l31 >> 4) << 1) + ((i16 & 3) >> 1)
            return (grp * 16 + r) * KP + ((chunk ^ rot4(r)) << 4) + 8 * (i16 & 1)
        assert conflicts(tr, TR_GROUPS, 8) == 0
    # the swizzle is a bijection of the 16 chunks of every row
    for r in range(64):
        assert sorted(c ^ rot4(r & 31) for c in range(16)) == list(range(16))


def test_attention32_accumulator_rows_match_the_transpose_reads():
    """register r of lane-half hi of a 32x32x16 accumulator is column-side row (r&3) + 8 (r>>2) + 4 hi; the two transpose reads of a
    16-row group fetch exactly rows {4hi..4hi+3} and {8+4hi..8+4hi+3}: the k-slot order of pack8(j) (registers 8j..8j+7)"""
    for hi in range(2):
        for j in range(2):
            rows = [(r & 3) + 8 * (r >> 2) + 4 * hi for r in range(8 * j, 8 * j + 8)]
            base = 16 * j
            assert rows == [base + 4 * hi + e for e in range(4)] + [base + 8 + 4 * hi + e for e in range(4)]


def test_gemm_pp_lds_image_staging_map_and_epilogue():
    """gemm_pp.hip (8-wave ping-pong GEMM): (1) LDS rows of 128 B with chunk ^ (row & 7): every fragment read of the wave's 8 A / 4 B blocks
    is conflict-free; (2) the direct-to-LDS piece (lane l -> LDS byte 16 l = row l>>3, position l&7, SOURCE chunk (l&7) ^ (l>>3)) and the
    fragment read apply the same involution: the reader of (row, chunk c) finds source chunk c; (3) the staging units V0, V1, V2 of the
    eight waves tile the 256 A rows and 256 B rows exactly once; (4) the v_permlane16_swap pairing of the epilogue gives every lane
    8 consecutive columns and the four lane rows tile the 32 columns of a tile pair"""
    for ks, blk in itertools.product(range(2), range(8)):
        assert conflicts(lambda lane: (blk * 16 + (lane & 15)) * 128 + ((((ks * 4) + (lane >> 4)) ^ (lane & 7)) << 4), B128_GROUPS, 16) == 0
    # (2) piece base rows are multiples of 8, so (row & 7) == l >> 3 inside a piece
    for base in range(0, 256, 8):
        for lane in range(64):
            row, pos = base + (lane >> 3), lane & 7
            src_chunk = pos ^ (lane >> 3)
            assert lane * 16 == (row - base) * 128 + pos * 16          # lane-linear LDS image of the piece
            assert (src_chunk ^ (row & 7)) == pos                       # a reader of chunk c looks at position c ^ (row & 7)
    # (3) units: wave w = (g, wc); A pieces rows g*128 + a*64 + wc*16 + 8 p (a, p in 0..1); B pieces rows w*32 + 8 p (p in 0..3)
    a_rows, b_rows = [], []
    for w in range(8):
        g, wc = w >> 2, w & 3
        for a_, p_ in itertools.product(range(2), range(2)):
            a_rows += [g * 128 + a_ * 64 + wc * 16 + 8 * p_ + r for r in range(8)]
        for p_ in range(4):
            b_rows += [w * 32 + 8 * p_ + r for r in range(8)]
    assert sorted(a_rows) == list(range(256)) and sorted(b_rows) == list(range(256))
    # unit V0 (a = 0) holds exactly the rows the a = 0 phases of both groups read: g*128 + 0..63
    v0 = sorted(g * 128 + wc * 16 + 8 * p_ + r for g in range(2) for wc in range(4) for p_ in range(2) for r in range(8))
    assert v0 == sorted(g * 128 + r for g in range(2) for r in range(64))
    # (4) lane row q (= lane >> 4) of column tiles (2 jj, 2 jj + 1): before the swap it holds columns 16 j + 4 q + e of tile j;
    # v_permlane16_swap exchanges the ODD rows of x (tile 2 jj) with the EVEN rows of y (tile 2 jj + 1)
    x = {q: [4 * q + e for e in range(4)] for q in range(4)}              # tile 2 jj
    y = {q: [16 + 4 * q + e for e in range(4)] for q in range(4)}         # tile 2 jj + 1
    x2, y2 = dict(x), dict(y)
    x2[1], y2[0] = y[0], x[1]
    x2[3], y2[2] = y[2], x[3]
    cols = []
    for q in range(4):
        mine = x2[q] + y2[q]
        start = 16 * (q & 1) + 8 * (q >> 1)                              # the store offset the kernel uses
        assert mine == list(range(start, start + 8))
        cols += mine
    assert sorted(cols) == list(range(32))


def test_gemm_pp_nn_operand_image():
    """gemm_pp.hip, NN form (dgrad c = s W from the stored weight): the B operand of a K tile is an LDS image [64 contraction rows][512 B =
    256 output columns], 16-byte chunk c of row r at position c ^ f(r), f(r) = 2 ((r & 3) + 4 ((r >> 3) & 1)).
    (1) staging: piece p of wave w = rows 8 w + 2 p + (l >> 5), lane l -> LDS slot l of the piece, SOURCE chunk (l & 31) ^ f(row): the
        eight waves' 32 pieces tile the image exactly once and a reader of (row, chunk c) finds source chunk c;
    (2) the MFMA operand of column tile j (8 consecutive contraction indices of one output column) is gathered by ds_read_b64_tr_b16 at
        row 8 hi + (i16 >> 2) (+4 for the second half, +32 for the second k-step), chunk (8 wc + 2 j + ((i16 & 3) >> 1)) ^ f(row),
        byte 8 (i16 & 1): conflict-free under the 32-lane groups of the transpose read, for every wave column wc, tile j, half, k-step"""
    f = lambda r: 2 * ((r & 3) + 4 * ((r >> 3) & 1))           # noqa: E731
    seen = {}
    for w, p_, lane in itertools.product(range(8), range(4), range(64)):
        row, pos = 8 * w + 2 * p_ + (lane >> 5), lane & 31
        lds = (8 * w + 2 * p_) * 512 + lane * 16                  # lane-linear 1-KiB piece at the piece's first row
        assert lds == row * 512 + pos * 16
        # the kernel's per-lane source offset uses r = 2 (p & 1) + (l >> 5) and (wave & 1): the same f as f(row)
        assert 2 * (((2 * (p_ & 1) + (lane >> 5)) & 3) + 4 * (w & 1)) == f(row)
        seen[(row, pos)] = pos ^ f(row)                          # source chunk stored at this position
    assert len(seen) == 64 * 32
    for row, c in itertools.product(range(64), range(32)):
        assert seen[(row, c ^ f(row))] == c
        assert 0 <= (c ^ f(row)) < 32
    for wc, j, half, ks in itertools.product(range(4), range(4), range(2), range(2)):
        def tr(lane, wc=wc, j=j, half=half, ks=ks):
            hi, i16 = lane >> 4, lane & 15
            row = 8 * hi + (i16 >> 2) + 4 * half + 32 * ks
            fl = 2 * (((i16 >> 2) & 3) + 4 * (hi & 1))         # what the kernel computes from the lane id alone
            assert fl == f(row)
            return row * 512 + (((8 * wc + 2 * j + ((i16 & 3) >> 1)) ^ fl) << 4) + 8 * (i16 & 1)
        assert conflicts(tr, TR_GROUPS, 8) == 0


def test_gemm_product_chunk_swizzle():
    """gemm.hip: LDS rows of 128 B, 16-byte chunk XOR (row & 7): fragment reads (row l&15, chunk kk*4 + l>>4) are conflict-free"""
    for kk in range(2):
        assert conflicts(lambda lane: (lane & 15) * 128 + ((((kk * 4) + (lane >> 4)) ^ (lane & 7)) << 4), B128_GROUPS, 16) == 0


def test_generic_attention_tile_swizzle():
    """attention.hip: row-major tiles of 128 / 256 / 512-byte rows, chunk ^ (row & (min(chunks, 16) - 1)) (swz_mask): the 16-row fragment
    reads of every 64-byte K chunk group are conflict-free"""
    for pitch in (128, 256, 512):
        nch = pitch // 16
        sw = min(nch, 16) - 1
        for kk in range(nch // 4):
            assert conflicts(lambda lane: (lane & 15) * pitch + ((((kk * 4) + (lane >> 4)) ^ ((lane & 15) & sw)) << 4), B128_GROUPS, 16) == 0


def test_linear_stream_lds_images_are_conflict_free():
    """linear_stream.hip (round 4).  Forward / dgrad operand images in the wave-private LDS rings:
    (1) dgrad W image [32 contraction rows][128 B], 16-byte chunk c of row r at c ^ g(r), g(r) = 2 (((r >> 1) & 1) + 2 ((r >> 3) & 1)): the staging
        map (piece p: rows 8 p + (l >> 3), slot l & 7, source chunk slot ^ g) tiles the image once, and the MFMA operand of column tile j (8
        consecutive contraction rows of one output column) gathered by ds_read_b64_tr_b16 at row 8 hi + 4 h + (i16 >> 2), chunk (2 j + ((i16 & 3) >> 1))
        ^ g, byte 8 (i16 & 1) is conflict-free under the 32-lane transpose-read groups;
    (2) dgrad s image [16 rows][64 B] per row block, chunk c of row r at c ^ ((-(r >> 2)) & 3): the 16-row ds_read_b128 fragments (row l & 15,
        chunk l >> 4) are conflict-free under the hardware's lane groups"""
    g = lambda r: 2 * (((r >> 1) & 1) + 2 * ((r >> 3) & 1))           # noqa: E731
    seen = {}
    for p_, lane in itertools.product(range(4), range(64)):
        row, slot = 8 * p_ + (lane >> 3), lane & 7
        assert 2 * (((lane >> 4) & 1) + 2 * (p_ & 1)) == g(row)       # what the kernel computes from the lane id and p & 1
        seen[(row, slot)] = slot ^ g(row)
    assert len(seen) == 32 * 8
    for row, c in itertools.product(range(32), range(8)):
        assert seen[(row, c ^ g(row))] == c
    for j, h in itertools.product(range(4), range(2)):
        def tr(lane, j=j, h=h):
            hi, i16 = lane >> 4, lane & 15
            row = 8 * hi + 4 * h + (i16 >> 2)
            gl = 2 * (((i16 >> 3) & 1) + 2 * (hi & 1))
            assert gl == g(row)
            a0 = (8 * hi + (i16 >> 2)) * 128 + ((((i16 & 3) >> 1) ^ gl) << 4) + 8 * (i16 & 1)      # the kernel's base (j = 0, h = 0)
            addr = (a0 ^ (j << 5)) + 512 * h
            assert addr == row * 128 + (((2 * j + ((i16 & 3) >> 1)) ^ g(row)) << 4) + 8 * (i16 & 1)
            return addr
        assert conflicts(tr, TR_GROUPS, 8) == 0
    f = lambda r: (-(r >> 2)) & 3                                     # noqa: E731
    img = {}
    for lane in range(64):
        row, slot = lane >> 2, lane & 3
        assert ((-(lane >> 4)) & 3) == f(row)
        img[(row, slot)] = slot ^ f(row)
    for row, c in itertools.product(range(16), range(4)):
        assert img[(row, c ^ f(row))] == c

    def sfrag(lane):
        hi, i16 = lane >> 4, lane & 15
        return i16 * 64 + ((hi ^ f(i16)) << 4)
    assert conflicts(sfrag, B128_GROUPS, 16) == 0


def test_gemma3_mm_row_intervals_equal_hf_mask_semantics():
    """engine_gemma3_mm.mm_row_intervals: per-row key intervals == (causal [and sliding window]) OR same-image-block, the mask HF's
    create_masks_for_vision_model builds from token_type_ids (image tokens of one block attend to each other in both directions)"""
    import torch
    import importlib.util
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "lrp-explains-transformers_amd", "engine_gemma3_mm.py")
    src = open(path).read()
    ns = {}
    start = src.index("def mm_row_intervals")
    exec("import torch\n" + src[start: src.index("def vision_weights_from_hf")], ns)       # the pure function alone (the module needs the .so)
    tt = torch.zeros(3, 40, dtype=torch.long)
    tt[0, 3:11] = 1; tt[0, 20:24] = 1; tt[1, 0:8] = 1; tt[2, 32:40] = 1
    for w in (5, 16, 64):
        iv = ns["mm_row_intervals"](tt, w)
        B, S = tt.shape
        i = torch.arange(S)
        for b in range(B):
            row = tt[b].bool()
            blk, cur = torch.full((S,), -1), -1
            for j in range(S):
                if row[j] and (j == 0 or not row[j - 1]):
                    cur += 1
                if row[j]:
                    blk[j] = cur
            same = (blk[:, None] == blk[None, :]) & (blk[:, None] >= 0)
            causal = i[None, :] <= i[:, None]
            for name, m in (("global", causal | same), ("local", (causal & (i[None, :] > i[:, None] - w)) | same)):
                lo, hi = iv[name]
                assert torch.equal(m, (i[None, :] >= lo[b][:, None]) & (i[None, :] < hi[b][:, None])), (b, name, w)
