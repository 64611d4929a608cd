"""CPU check of the LDS stages of the X pipeline (pafuse_amd/csrc/xgemm.hpp), of the large-tile weight-gradient kernel
(train_kernels.hpp, check_tn_tile) and of the strip kernel of the plain bf16x3 layers (sgemm.hpp, check_strip_tile): no GPU needed.

For every tile shape the library launches it replays, in integers, (a) the LDS-DMA fill of one stage - which source bytes
(row, slice, sub-block of 8 k) each lane of each 1 KiB wave instruction deposits where - and (b) the fragment reads of the K
loop - which LDS bytes lane (r, h) of wave (wm, wn) reads for 16-deep step s2, column block nt, slice sl - and asserts
  1. every fragment read finds the sub-block the MFMA expects there: row = the lane's tile row, k = 16 s2 + 8 h .. + 7;
  2. every ds_read_b128 is conflict-free under the gfx950 rule (MI355X_MICROARCH.md, LDS: a b128 wave access is served in
     four groups of 16 lanes, 64 banks of 4 bytes, one cycle per group when no two lanes of a group touch the same bank
     at different addresses).

    python tools/lds_layout_check.py
"""
import itertools

B128_GROUPS = [
    list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)),
    list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32)),
    list(range(32, 36)) + list(range(44, 48)) + list(range(52, 60)),
    list(range(36, 44)) + list(range(48, 52)) + list(range(60, 64)),
]


def swizzle(bkc, row):
    return (row >> 2) & 3 if bkc == 32 else (row >> 3) & 1


def check_tile(wm_, wn_, nt_, bkc):
    nw = wm_ * wn_
    bm, bn = wm_ * 32, wn_ * nt_ * 32
    prowb = bkc * 2
    rpi, cpr = 1024 // prowb, prowb // 16
    a_plane, w_plane = bm * prowb, bn * prowb
    iap, iwp = a_plane // 1024, w_plane // 1024
    ia, iw = 3 * iap, 3 * iwp
    a_bytes = 3 * a_plane
    # ---- (a) the fill: LDS 16-byte slot -> (operand, row, slice, sub-block of the chunk)
    lds = {}
    for i in range(ia + iw):
        is_a = i < ia
        plane = i // iap if is_a else (i - ia) // iwp
        for lane in range(64):
            row = rpi * (i % iap if is_a else (i - ia) % iwp) + lane // cpr
            sb = (lane % cpr) ^ swizzle(bkc, row)
            addr = i * 1024 + lane * 16
            assert addr not in lds
            lds[addr] = ("A" if is_a else "W", row, plane, sb)
    assert len(lds) == (ia + iw) * 64
    # ---- (b) the reads
    ns2 = bkc // 16
    for wm, wn in itertools.product(range(wm_), range(wn_)):
        for s2 in range(ns2):
            for sl in range(3):
                for which, nts in (("A", [0]), ("W", range(nt_))):
                    for nt in nts:
                        addrs = []
                        for lane in range(64):
                            r, h = lane & 31, lane >> 5
                            pos = (((2 * s2 + h) ^ swizzle(bkc, r)) & (cpr - 1)) * 16
                            if which == "A":
                                row = wm * 32 + r
                                addr = row * prowb + pos + sl * a_plane
                            else:
                                row = wn * nt_ * 32 + nt * 32 + r
                                addr = a_bytes + (wn * nt_ * 32 + r) * prowb + pos + nt * 32 * prowb + sl * w_plane
                            got = lds[addr]
                            assert got == (which, row, sl, 2 * s2 + h), (which, lane, got, (row, sl, 2 * s2 + h))
                            addrs.append(addr)
                        for grp in B128_GROUPS:      # a b128 access of a lane covers banks a/4 .. a/4 + 3 (mod 64)
                            seen = {}
                            for lane in grp:
                                for d in range(4):
                                    bank = (addrs[lane] // 4 + d) % 64
                                    assert seen.setdefault(bank, addrs[lane]) == addrs[lane], ("bank conflict", which, wm, wn, s2, sl, nt)
    return (ia + iw), a_bytes + 3 * w_plane


def check_tn_tile(nb_, kb_, wn_, wk_):
    """tn_split_big_kernel<NB, KB, WN, WK> (pafuse_amd/csrc/train_kernels.hpp): one LDS stage of the weight-gradient kernel.
    Fill: thread t stages 8 rows of the contraction x 2 columns of Y (t < 32 NB) or X and writes, per column and slice, one
    16-byte slot at slot(pitch, m group, column).  Reads: lane (r, h) of wave (wn, wk) takes m group h of column 32 block + r.
    Asserts that every read finds (operand, slice, m group, column) it expects and that reads and writes are conflict-free
    under the b128 group rule."""
    nthr = 64 * wn_ * wk_
    assert 32 * (nb_ + kb_) == nthr
    yc, xc = 32 * nb_, 32 * kb_
    y_plane, x_plane = 2 * yc * 16, 2 * xc * 16
    x_base = 3 * y_plane

    def slot(pitch, mg, col):
        return (mg * pitch + (col & ~15) + ((col & 15) ^ ((col >> 4) & 1))) * 16

    lds = {}
    writes = []   # per (j, slice): address of every thread
    for j in range(2):
        for sl in range(3):
            addrs = []
            for t in range(nthr):
                is_y = t < 32 * nb_
                st = t if is_y else t - 32 * nb_
                pairs = 16 * nb_ if is_y else 16 * kb_
                smg, sc2 = st // pairs, 2 * (st % pairs)
                a = (slot(yc, smg, sc2 + j) if is_y else x_base + slot(xc, smg, sc2 + j)) + sl * (y_plane if is_y else x_plane)
                assert a not in lds
                lds[a] = ("Y" if is_y else "X", sl, smg, sc2 + j)
                addrs.append(a)
            writes.append(addrs)
    assert len(lds) == 3 * 2 * (yc + xc)
    bnw, bkw = nb_ // wn_, kb_ // wk_
    for wave in range(wn_ * wk_):
        wn, wk = wave // wk_, wave % wk_
        for sl in range(3):
            for which, blocks in (("Y", [wn * bnw + a for a in range(bnw)]), ("X", [wk * bkw + b for b in range(bkw)])):
                for blk in blocks:
                    addrs = []
                    for lane in range(64):
                        r, h = lane & 31, lane >> 5
                        col = blk * 32 + r
                        a = (slot(yc, h, col) + sl * y_plane) if which == "Y" else (x_base + slot(xc, h, col) + sl * x_plane)
                        assert lds[a] == (which, sl, h, col), (which, lane, lds[a], (sl, h, col))
                        addrs.append(a)
                    _assert_b128_conflict_free(addrs, ("read", which, wave, sl, blk))
    for w, addrs in enumerate(writes):     # the stage writes, wave by wave
        for wave in range(nthr // 64):
            _assert_b128_conflict_free(addrs[64 * wave:64 * wave + 64], ("write", w, wave))
    return 3 * 2 * (yc + xc) * 16


def _assert_b128_conflict_free(addrs, what):
    for grp in B128_GROUPS:
        seen = {}
        for lane in grp:
            for d in range(4):
                bank = (addrs[lane] // 4 + d) % 64
                assert seen.setdefault(bank, addrs[lane]) == addrs[lane], ("bank conflict",) + tuple(what)


def strip_f(x):
    """position swizzle of the strip kernel's A stage (pafuse_amd/csrc/sgemm.hpp strip_f): x = (row >> 1) & 7"""
    return x ^ ((((x + 2) >> 2) & 1) << 1)


def check_strip_tile(nb_, rg_, nw_):
    """sgemm2_kernel<NB, RG, NW> (pafuse_amd/csrc/sgemm.hpp): one stage of the strip kernel - the waves' PRIVATE A rows (fp32, 128 bytes
    per row of a 32-deep chunk, RG groups of 16 rows per wave, filled by RG * 2 DMA instructions of that wave with the swizzle on the
    source address) and the W' tile in its M16 image layout (rows of 192 bytes, sub-block sb at ((sb + (n >> 1)) & 3) * 48, slices at
    + 0 / 16 / 32, copied as it lies).  Asserts that lane (c, qd)'s two A reads find row c's 16-byte chunks 2 qd and 2 qd + 1, that its
    three W' reads find (column 16 nb + c, sub-block qd, slice), and that every ds_read_b128 is conflict-free."""
    a_wave = rg_ * 2048
    a_bytes = nw_ * a_wave
    bn = 16 * nb_
    lds = {}
    for wave in range(nw_):
        for i in range(rg_ * 2):                       # A instruction i of the wave: its rows 8 i .. 8 i + 7
            for lane in range(64):
                row = 8 * i + (lane >> 3)
                x = (4 * (i & 1) + (lane >> 4)) & 7
                assert x == (row >> 1) & 7
                ch = (lane & 7) ^ strip_f(x)
                addr = wave * a_wave + i * 1024 + lane * 16
                assert addr not in lds
                lds[addr] = ("A", wave, row, ch)
    for n in range(bn):                                # the W' tile's chunk of the image, as it lies
        for sb in range(4):
            for sl in range(3):
                addr = a_bytes + n * 192 + ((sb + (n >> 1)) & 3) * 48 + 16 * sl
                assert addr not in lds
                lds[addr] = ("W", n, sb, sl)
    assert len(lds) * 16 == a_bytes + bn * 192
    for wave in range(nw_):
        for g in range(rg_):
            for o in range(2):
                addrs = []
                for lane in range(64):
                    c, qd = lane & 15, lane >> 4
                    fc = strip_f((c >> 1) & 7)
                    a0 = wave * a_wave + c * 128 + ((2 * qd) ^ fc) * 16
                    addr = (a0 ^ 16 if o else a0) + g * 2048
                    assert lds[addr] == ("A", wave, 16 * g + c, 2 * qd + o), (lane, lds[addr], (16 * g + c, 2 * qd + o))
                    addrs.append(addr)
                _assert_b128_conflict_free(addrs, ("strip A", wave, g, o))
        for nb in range(nb_):
            for sl in range(3):
                addrs = []
                for lane in range(64):
                    c, qd = lane & 15, lane >> 4
                    addr = a_bytes + c * 192 + ((qd + (c >> 1)) & 3) * 48 + nb * 16 * 192 + 16 * sl
                    assert lds[addr] == ("W", 16 * nb + c, qd, sl), (lane, lds[addr])
                    addrs.append(addr)
                _assert_b128_conflict_free(addrs, ("strip W'", wave, nb, sl))
    return a_bytes + bn * 192


STRIP_SHAPES = [(8, 2, 4), (7, 2, 4), (6, 2, 4)]
X_SHAPES = [(4, 2, 2, 16), (4, 1, 7, 16), (4, 1, 3, 16), (4, 2, 6, 16), (2, 2, 4, 16), (4, 1, 9, 16), (4, 1, 6, 16), (5, 1, 3, 16),
            (4, 2, 2, 32), (2, 2, 4, 32), (4, 1, 7, 32)]
TN_SHAPES = [(8, 8, 4, 2), (12, 4, 4, 2), (6, 6, 3, 2), (7, 7, 7, 1)]

if __name__ == "__main__":
    for s in STRIP_SHAPES:
        print(f"sgemm2_kernel<NB={s[0]}, RG={s[1]}, NW={s[2]}>: stage {check_strip_tile(*s)} bytes: fill = reads, no bank conflicts")
    for s in TN_SHAPES:
        print(f"tn_split_big_kernel<NB={s[0]}, KB={s[1]}, WN={s[2]}, WK={s[3]}>: stage {check_tn_tile(*s)} bytes: fill = reads, no bank conflicts")
    for s in X_SHAPES:
        pieces, stage = check_tile(*s)
        print(f"XTile<WM={s[0]}, WN={s[1]}, NT={s[2]}, BKC={s[3]}>: {pieces} DMA pieces, stage {stage} bytes: fill = reads, no bank conflicts")
