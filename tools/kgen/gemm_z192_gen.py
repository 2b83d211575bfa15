"""Generator of the hand-scheduled persistent 256(M) x 192(N) x 64 f16 GEMM (sam_road_amd/csrc/gemm_z192.hip):
        OUT16[M,N] = act(A[M,K] W[N,K]^T + bias)         fp16 out, fp32 accumulate
for the four big linear layers of a SAM ViT block (reference model.py:245-258 via segment_anything's Block).

    python tools/kgen/gemm_z192_gen.py            # writes sam_road_amd/csrc/gemm_z192_body_act{0,1}.inc

Organisation (why: DESIGN.md §4.1; measured in profiles/r04_feedx_probe*.txt):
  * ONE wave per SIMD (4 waves, 2 x 2), wave tile 128(M) x 96(N) = 12 accumulator tiles of v_mfma_f32_32x32x16_f16 in
    a[0:191] ("transposed": A operand = weight rows, so a lane owns 4 consecutive output columns per register quad);
    fragments double-buffered in a[192:247], read ONE k-step ahead by ds_read_b128 placed one per MFMA gap;
  * operand k-tiles HBM/L2 -> LDS by buffer_load ... lds (1 KiB pieces of 8 rows x 128 B, XOR swizzle on the source
    address), ONE piece per MFMA gap, never a burst: W in a two-slot ring, X in a three-slot ring (144 KiB), so every
    piece is issued at least ~one k-tile (~0.9 us) before the barrier that publishes it;
  * one s_barrier per k-tile, issued right AFTER the first MFMA of the k-tile's last k-step (the matrix pipe has work
    while the waves rendezvous); counted s_waitcnt vmcnt(N), never 0 in the loop;
  * a tile's accumulators leave through v[64:255] (v_accvgpr_read at the tile end, the next tile starts with C = 0);
    bias / activation / fp16 packing / v_permlane32_swap -> 16-byte stores of that finished tile ride in the MFMA gaps of
    the NEXT tile's k-loop (deferred epilogue), only a workgroup's last tile has an exposed one.
The code is emitted through tools/kgen/asmdsl.py, whose emulator checks it on the CPU (tests/test_kgen_emulator.py).
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from asmdsl import A, EXEC_HI, EXEC_LO, M0, Prog, S, V, Workgroup, check_hazards, neg  # noqa: E402

# ---------------------------------------------------------------------------------------------- fixed register map
KARG, BID, WAVE = S(36, 2), S(38), S(39)
RS_X, RS_W, RS_O, RS_B = S(40, 4), S(44, 4), S(48, 4), S(52, 4)
CUR, NXT = S(56, 4), S(60, 4)            # tile table entries {x_off, w_off, out_off, bias_off} (bytes)
LDA, LDW, LDC, NTILES = S(64), S(65), S(66), S(67)
NKB, GRID, A_BYTES, W_BYTES = S(68), S(69), S(70), S(71)
O_BYTES, B_BYTES, FLAGS, PAD0 = S(72), S(73), S(74), S(75)
TABLE = S(76, 2)
DW, DWL, DX, DXL = S(78), S(79), S(80), S(81)       # DMA streams: soffset of the stream's k-tile, k-tiles left in its tile
NK, WLDS = S(82), S(83)
TW, TX = S(84), S(85)                     # running piece soffsets
LDW32, LDA32, LDC32 = S(86), S(87), S(88)
TLEFT, CIDX = S(89), S(90)               # tiles left (incl. current), table byte offset of CUR's entry
PO, PB = S(91), S(92)                     # finished tile: out offset, bias LDS slot offset (0 / 768)
CB = S(93)                                # current tile's bias slot offset
KBL = S(94)                               # bodies left in the current tile
T0, T1, T2, T3 = S(95), S(96), S(97), S(98)
HAVEP, G16 = S(99), S(75)          # G16 reuses the padding dword of the last kernarg quad

TID, LANE, VW, VX = V(4), V(5), V(6), V(7)
WF = V(8, 4)                              # W fragment LDS address per k-step
XF = V(12, 12)                            # X fragment LDS address per (slot, k-step)
VO, VB, VBD, VC0, VBL = V(24), V(25), V(26), V(27), V(28)
TMP = V(30, 34)                           # v30..v63 (even base: register tuples must be 64-bit aligned on gfx90a+)
SETB = V(64, 192)
ACC = A(0, 192)


def FRW(set_, i):
    return A(192 + 28 * set_ + 4 * i, 4)


def FRX(set_, j):
    return A(192 + 28 * set_ + 12 + 4 * j, 4)


W_SLOT, X_BASE, X_SLOT, BIAS_LDS = 24576, 49152, 32768, 147456
LDS_BYTES = BIAS_LDS + 2 * 768

# kernarg layout (struct ZParams in gemm_z192.hip)
K_A, K_W, K_BIAS, K_OUT, K_TABLE = 0, 8, 16, 24, 32
K_DIMS, K_DIMS2, K_DIMS3 = 48, 64, 80
KARG_BYTES = 96

GELU_C = [0.00048291164585022967, -0.0071898452371611365, 0.05218537922649359, 0.4595148493607732, 1.1510354141727006, 1.0]


class ZGen:
    def __init__(self, act=0, deferred=True, sched=None):
        self.act, self.deferred = act, deferred
        self.p = Prog()
        self.sched = sched or {}

    # ------------------------------------------------------------------------------------------ prologue
    def prologue(self):
        p = self.p
        p.raw("s_mov_b64 s[36:37], %0")
        p.raw("s_mov_b32 s38, %1")
        p.raw("v_mov_b32 v4, %2")
        p.s_load_dwordx(2, RS_X.sub(0, 2), KARG, K_A)
        p.s_load_dwordx(2, RS_W.sub(0, 2), KARG, K_W)
        p.s_load_dwordx(2, RS_B.sub(0, 2), KARG, K_BIAS)
        p.s_load_dwordx(2, RS_O.sub(0, 2), KARG, K_OUT)
        p.s_load_dwordx(2, TABLE, KARG, K_TABLE)
        p.s_load_dwordx(4, S(64, 4), KARG, K_DIMS)
        p.s_load_dwordx(4, S(68, 4), KARG, K_DIMS2)
        p.s_load_dwordx(4, S(72, 4), KARG, K_DIMS3)
        # lane constants meanwhile
        p.v_and_b32(LANE, 63, TID)
        p.v_lshrrev_b32(TMP[0], 6, TID)
        p.s_nop(0)
        p.v_readfirstlane_b32(WAVE, TMP[0])
        p.s_waitcnt(lgkmcnt=0)
        for rs, nbytes in ((RS_X, A_BYTES), (RS_W, W_BYTES), (RS_O, O_BYTES), (RS_B, B_BYTES)):
            p.s_and_b32(rs[1], rs[1], 0xFFFF)
            p.s_mov_b32(rs[2], nbytes)
            p.s_mov_b32(rs[3], 0x00020000)
        p.s_lshl_b32(WLDS, WAVE, 10)
        p.s_lshl_b32(LDW32, LDW, 5)
        p.s_lshl_b32(LDA32, LDA, 5)
        p.s_lshl_b32(LDC32, LDC, 5)
        p.s_mul_i32(NK, NKB, 12)
        # DMA lane offsets: row lr = wave*8 + lane/8, chunk (lane&7) ^ ((lr>>1)&7)
        t0, t1, t2, t3 = TMP[0], TMP[1], TMP[2], TMP[3]
        p.v_lshrrev_b32(t0, 3, LANE)
        p.v_lshl_add_u32(t0, WAVE, 3, t0)                 # lr
        p.v_lshrrev_b32(t1, 1, t0)
        p.v_and_b32(t1, 7, t1)
        p.v_and_b32(t2, 7, LANE)
        p.v_xor_b32(t1, t1, t2)                           # swizzled chunk
        p.v_mul_lo_u32(t2, t0, LDW)
        p.v_lshl_add_u32(VW, t1, 4, t2)
        p.v_mul_lo_u32(t2, t0, LDA)
        p.v_lshl_add_u32(VX, t1, 4, t2)
        # fragment addresses: frow = lane&31, fhalf = lane>>5, key = (frow>>1)&7
        p.v_and_b32(t0, 31, LANE)                         # frow
        p.v_lshrrev_b32(t1, 5, LANE)                      # fhalf
        p.v_lshrrev_b32(t2, 1, t0)
        p.v_and_b32(t2, 7, t2)                            # key
        p.s_and_b32(T0, WAVE, 1)                          # wn
        p.s_lshr_b32(T1, WAVE, 1)                         # wm
        p.s_mul_i32(T2, T0, 96 * 128)
        p.s_mul_i32(T3, T1, 128 * 128)
        p.v_lshlrev_b32(t3, 7, t0)                        # frow * 128
        for ks in range(4):
            c = TMP[4]
            p.v_or_b32(c, 2 * ks, t1)                     # 2 ks + fhalf   (2 ks is even, fhalf is bit 0)
            p.v_xor_b32(c, c, t2)
            p.v_lshl_add_u32(c, c, 4, t3)                 # ((2ks+fhalf)^key)*16 + frow*128
            p.v_add_u32(WF[ks], T2, c)
            for slot in range(3):
                p.v_add_u32(XF[slot * 4 + ks], T3, c)
                p.v_add_u32(XF[slot * 4 + ks], X_BASE + slot * X_SLOT, XF[slot * 4 + ks])
        # output lane offset: (wm*128 + frow) * ldc2 + wn*192 + fhalf*16
        p.s_lshl_b32(T3, T1, 7)
        p.v_add_u32(t3, T3, t0)
        p.v_mul_lo_u32(t3, t3, LDC)
        p.s_mul_i32(T2, T0, 192)
        p.v_lshl_add_u32(t3, t1, 4, t3)
        p.v_add_u32(VO, T2, t3)
        # bias: LDS read address (wn*96 + 4 fhalf) floats, DMA source offset wave*192 + lane*16
        p.s_mul_i32(T2, T0, 384)
        p.v_lshlrev_b32(t3, 4, t1)
        p.v_add_u32(VBL, T2, t3)                          # lane part; VB = VBL + BIAS_LDS + slot offset of the finished tile
        p.v_add_u32(VB, BIAS_LDS, VBL)
        p.s_mul_i32(T2, WAVE, 192)
        p.v_lshlrev_b32(t3, 4, LANE)
        p.v_add_u32(VBD, T2, t3)
        p.v_mov_b32(VC0, float(GELU_C[0]))
        # tiles of this workgroup: idx = bid, bid + grid, ...; my_tiles = ceil((ntiles - bid) / grid)   (bid < ntiles)
        p.s_sub_u32(T0, NTILES, BID)
        p.s_add_u32(T0, T0, GRID)
        p.s_sub_u32(T0, T0, 1)
        # T0 / GRID by repeated subtraction (at most a handful of tiles per workgroup)
        p.s_mov_b32(TLEFT, 0)
        lp = p.newlabel("cnt")
        p.label(lp)
        p.s_add_u32(TLEFT, TLEFT, 1)
        p.s_sub_u32(T0, T0, GRID)
        p.s_cmp_ge_u32(T0, GRID)
        p.s_cbranch_scc1(lp)
        p.s_lshl_b32(CIDX, BID, 4)
        p.s_lshl_b32(G16, GRID, 4)
        p.s_load_dwordx(4, CUR, TABLE, CIDX)
        self.load_next()
        p.s_waitcnt(lgkmcnt=0)
        p.s_mov_b32(CB, 0)
        p.s_mov_b32(PB, 0)
        p.s_mov_b32(HAVEP, 0)
        p.s_mov_b32(DW, CUR[1])
        p.s_mov_b32(DX, CUR[0])
        p.s_mov_b32(DWL, NK)
        p.s_mov_b32(DXL, NK)
        # pipeline fill: W(0) X(0) X(1) W(1) pieces 0..4   (W(1) piece 5 and X(2) ride on k-tile 0)
        self.bias_dma()
        for it in self.w_pieces(0, range(6), True):
            it()
        for it in self.x_pieces(0, range(8), True):
            it()
        for it in self.x_pieces(1, range(8), True):
            it()
        for it in self.w_pieces(1, range(5), False):
            it()
        p.s_waitcnt(vmcnt=13)
        p.s_barrier()
        for it in self.frag_reads(0, 0, 0, 0):
            it()

    def load_next(self):
        """NXT <- table[NIDX] if another tile follows the one after CUR, else NXT <- CUR-like valid entry (re-fetch, never used)."""
        p = self.p
        p.s_add_u32(T2, CIDX, G16)
        p.s_cmp_gt_u32(TLEFT, 1)
        p.s_cselect_b32(T2, T2, CIDX)      # no tile after CUR: NXT = CUR (the streams re-fetch valid data nobody uses)
        p.s_load_dwordx(4, NXT, TABLE, T2)

    # ------------------------------------------------------------------------------------------ DMA pieces
    def w_pieces(self, slot, qs, advance):
        """Emitters (one per MFMA gap) for pieces qs of the W stream's current k-tile into W slot `slot`."""
        p = self.p
        out = []
        for q in qs:
            def emit(q=q):
                p.s_add_u32(M0, WLDS, slot * W_SLOT + 4096 * q)
                if q == 0:
                    p.s_mov_b32(TW, DW)
                else:
                    p.s_add_u32(TW, TW, LDW32)
                p.buffer_load_lds(16, VW, RS_W, TW)
                self.vm_log.append("W")
                if q == 5 and advance:
                    self.advance(DW, DWL, NXT[1])
            out.append(emit)
        return out

    def x_pieces(self, slot, qs, advance):
        p = self.p
        out = []
        for q in qs:
            def emit(q=q):
                p.s_add_u32(M0, WLDS, X_BASE + slot * X_SLOT + 4096 * q)
                if q == 0:
                    p.s_mov_b32(TX, DX)
                else:
                    p.s_add_u32(TX, TX, LDA32)
                p.buffer_load_lds(16, VX, RS_X, TX)
                self.vm_log.append("X")
                if q == 7 and advance:
                    self.advance(DX, DXL, NXT[0])
            out.append(emit)
        return out

    def advance(self, d, left, nxt_off):
        p = self.p
        p.s_add_u32(d, d, 128)
        p.s_sub_u32(left, left, 1)
        p.s_cmp_eq_u32(left, 0)
        p.s_cselect_b32(d, nxt_off, d)
        p.s_cselect_b32(left, NK, left)

    def bias_dma(self):
        """This tile's 192 bias floats -> LDS slot CB (12 lanes x 16 B per wave)."""
        p = self.p
        p.s_mov_b32(EXEC_LO, 0xFFF)
        p.s_mov_b32(EXEC_HI, 0)
        p.s_mul_i32(T3, WAVE, 192)
        p.s_add_u32(T3, T3, CB)
        p.s_add_u32(M0, T3, BIAS_LDS)
        p.s_nop(0)
        p.buffer_load_lds(16, VBD, RS_B, CUR[3])
        p.s_mov_b32(EXEC_LO, -1)
        p.s_mov_b32(EXEC_HI, -1)
        self.vm_log.append("B")

    # ------------------------------------------------------------------------------------------ fragments
    def frag_reads(self, set_, wslot, xslot, ks):
        """7 emitters: the fragments of k-step ks of the stage in (wslot, xslot) into fragment set set_.
        Order = order of first use by the MFMAs (i outer, j inner): X0 W0 X1 X2 X3 W1 W2."""
        p = self.p

        def rw(i):
            return lambda: (p.ds_read_b128(FRW(set_, i), WF[ks], wslot * W_SLOT + i * 4096), self.lg_log.append("F"))

        def rx(j):
            return lambda: (p.ds_read_b128(FRX(set_, j), XF[xslot * 4 + ks], j * 4096), self.lg_log.append("F"))
        return [rx(0), rw(0), rx(1), rx(2), rx(3), rw(1), rw(2)]

    # ------------------------------------------------------------------------------------------ epilogue pieces
    def epi_items(self, t):
        """The finished tile's accumulator tile t = 4 i + j (f32 in SETB[16t:16t+16]) -> bias, activation, fp16 pack,
        half-wave exchange, two 16-byte stores.  Returned as a list of small emitters (each a few instructions) that the
        caller weaves into MFMA gaps — or runs back to back for the exposed epilogue of a workgroup's last tile."""
        p = self.p
        i, j = t // 4, t % 4
        B = SETB.sub(16 * t, 16)
        bq = [TMP.sub(4 * q, 4) for q in range(4)]          # v28..v43: the 16 bias values of this accumulator tile
        tmp = [[TMP[16 + 3 * e + k] for k in range(3)] for e in range(4)]     # GELU temporaries of 4 elements in flight
        items = []

        def bias_reads():
            for q in range(4):
                p.ds_read_b128(bq[q], VB, i * 128 + q * 32)
                self.lg_log.append("E")
        items.append(("lds4", bias_reads))

        def wait_bias():
            # every LDS op issued after the 4 bias reads may stay in flight
            if "E" not in self.lg_log:
                return
            n = 0
            for k in reversed(self.lg_log):
                if k == "E":
                    break
                n += 1
            assert n <= 15
            p.s_waitcnt(lgkmcnt=n)
            self.lg_log = self.lg_log[len(self.lg_log) - n:]
        items.append(("wait", wait_bias))
        for r0 in range(0, 16, 4):
            def addb(r0=r0):
                for r in range(r0, r0 + 4):
                    p.v_add_f32(B[r], B[r], bq[r // 4][r % 4])
            items.append(("valu4", addb))
        if self.act == 1:
            for r0 in range(0, 16, 4):
                # 9 VALU per element, 4 elements interleaved for ILP
                def g1(r0=r0):
                    for e in range(4):
                        p.v_and_b32(tmp[e][0], 0x7FFFFFFF, B[r0 + e])                 # a = |x|
                    for e in range(4):
                        p.v_fmaak_f32(tmp[e][1], VC0, tmp[e][0], GELU_C[1])
                items.append(("valu8", g1))
                for c in (2, 3):
                    def g2(r0=r0, c=c):
                        for e in range(4):
                            p.v_fmaak_f32(tmp[e][1], tmp[e][1], tmp[e][0], GELU_C[c])
                    items.append(("valu4", g2))

                def g3(r0=r0):
                    for e in range(4):
                        p.v_fmaak_f32(tmp[e][1], tmp[e][1], tmp[e][0], GELU_C[4])
                    for e in range(4):
                        p.v_fmaak_f32(tmp[e][1], tmp[e][1], tmp[e][0], GELU_C[5])
                items.append(("valu8", g3))

                def g4(r0=r0):
                    for e in range(4):
                        p.v_exp_f32(tmp[e][2], neg(tmp[e][1]))                         # 2^-r
                    for e in range(4):
                        p.v_max_f32(tmp[e][1], 0, B[r0 + e])                           # max(x, 0)
                items.append(("valu8", g4))

                def g5(r0=r0):
                    for e in range(4):
                        p.v_fma_f32(B[r0 + e], neg(tmp[e][0]), tmp[e][2], tmp[e][1])    # max(x,0) - a 2^-r
                items.append(("valu4", g5))
        elif self.act == 2:
            for r0 in range(0, 16, 4):
                def relu(r0=r0):
                    for r in range(r0, r0 + 4):
                        p.v_max_f32(B[r], 0, B[r])
                items.append(("valu4", relu))

        def pack():
            for q in range(4):
                for d in range(2):
                    p.v_cvt_pk_f16_f32(B[2 * q + d], B[4 * q + 2 * d], B[4 * q + 2 * d + 1])
        items.append(("valu8", pack))

        def swap():
            p.s_nop(1)
            for kp in range(2):
                for d in range(2):
                    p.v_permlane32_swap_b32(B[4 * kp + d], B[4 * kp + 2 + d])
            # store soffset: out_off + j*32 rows + (i*32 + kp*16) columns
            if j == 0:
                p.s_add_u32(T2, PO, i * 64)
            else:
                p.s_add_u32(T2, T2, LDC32)
            p.s_add_u32(T3, T2, 32)
        items.append(("valu4", swap))

        def store0():
            p.buffer_store_dwordx4(B.sub(0, 4), VO, RS_O, T2)
            self.vm_log.append("S")
        items.append(("vmem", store0))

        def store1():
            p.buffer_store_dwordx4(B.sub(4, 4), VO, RS_O, T3)
            self.vm_log.append("S")
        items.append(("vmem", store1))
        return items

    # ------------------------------------------------------------------------------------------ one k-tile
    def ktile(self, kk, first, epi):
        """k-tile kk (0..11) of a 12-k-tile body.  first: the tile's first k-tile (k-step 0 starts the accumulators with C = 0).
        epi: list of epilogue emitters (kind, fn) to weave into this k-tile's gaps (consumed front to back)."""
        p = self.p
        ws, xs = kk % 2, kk % 3
        nws, nxs = (kk + 1) % 2, (kk + 1) % 3
        # DMA gap queue for this k-tile, in time order: k-steps 0..2 carry W(kk+1) piece 5 and X(kk+2) (slot (kk+2)%3);
        # k-step 3 (after the barrier) carries W(kk+2) pieces 0..4 into the slot k-tile kk has just retired.
        dma_012 = self.w_pieces(nws, [5], True) + self.x_pieces((kk + 2) % 3, range(8), True)
        dma_3 = self.w_pieces(ws, range(5), False)
        ngap = self.sched.get("dma_gaps", [7, 8, 9, 10, 11])
        xgaps = self.sched.get("x_gaps")        # optional explicit (ks, gap) list for the 9 pieces of k-steps 0..2
        if xgaps is None:
            xgaps = [(0, 7), (0, 8), (0, 10), (1, 7), (1, 9), (1, 11), (2, 7), (2, 9), (2, 11)]
        assert len(xgaps) == 9
        dma_at = {xgaps[n]: dma_012[n] for n in range(9)}
        for n in range(5):
            dma_at[(3, ngap[n])] = dma_3[n]
        for ks in range(4):
            set_ = ks & 1
            if ks < 3:
                reads = self.frag_reads(set_ ^ 1, ws, xs, ks + 1)
            else:
                reads = self.frag_reads(0, nws, nxs, 0)
            for m in range(12):
                i, j = m // 4, m % 4
                acc = ACC.sub(16 * (4 * i + j), 16)
                # operands of this k-step must have landed: frag reads were issued one k-step ago
                if m == 0:
                    self.wait_frags()
                p.v_mfma_f32_32x32x16_f16(acc, FRW(set_, i), FRX(set_, j), 0 if (first and ks == 0) else acc)
                if ks == 3 and m == 0:
                    self.barrier_point()
                if m < 7:
                    reads[m]()
                if (ks, m) in dma_at:
                    dma_at[(ks, m)]()
                # epilogue work: fill what is left of the gap
                budget = self.sched.get("epi_per_gap", 1)
                while epi and budget > 0:
                    kind, fn = epi[0]
                    if kind == "vmem" and ((ks, m) in dma_at):
                        break                       # at most one VMEM instruction per gap
                    if kind == "lds4" and m < 7:
                        break
                    epi.pop(0)
                    fn()
                    budget -= 1

    def wait_frags(self):
        """Before a k-step's first MFMA: the 7 fragment reads issued during the previous k-step have returned (LDS ops
        return in order: count what was issued after the last of them)."""
        if "F" not in self.lg_log:
            return
        n = 0
        for k in reversed(self.lg_log):
            if k == "F":
                break
            n += 1
        assert n <= 15
        self.p.s_waitcnt(lgkmcnt=n)
        self.lg_log = self.lg_log[len(self.lg_log) - n:]

    def barrier_point(self):
        """k-tile kk's hand-over (after the first MFMA of its last k-step): this wave's pieces of W(kk+1) and X(kk+1) have
        landed — everything issued after W(kk+1) piece 5 may stay in flight — and its reads of stage kk have returned."""
        p = self.p
        n = 0
        for k in reversed(self.vm_log):
            if k == "W":
                break
            n += 1
        assert n < 40
        p.s_waitcnt(vmcnt=n, lgkmcnt=0)
        self.lg_log = []
        p.s_barrier()

    # ------------------------------------------------------------------------------------------ body / tile / kernel
    def body(self, first, with_epi):
        epi = []
        if with_epi:
            for t in range(12):
                epi += self.epi_items(t)
        self.epi_total = len(epi)
        # canonical LDS-queue state at a body's entry (prologue and every body end leave exactly this): the 7 fragment reads of
        # the coming k-step are the youngest LDS operations
        self.lg_log = ["F"] * 7
        for kk in range(12):
            self.ktile(kk, first and kk == 0, epi)
        assert not epi, f"{len(epi)} epilogue items did not fit the body"
        assert self.lg_log[-7:] == ["F"] * 7 and "E" not in self.lg_log, "a body must end with its 7 fragment reads youngest"

    def tile_end(self):
        """Accumulators -> v[64:255] (the next tile starts with C = 0)."""
        p = self.p
        p.s_nop(7)
        p.s_nop(7)
        for r in range(192):
            p.v_accvgpr_read_b32(SETB[r], ACC[r])

    def exposed_epilogue(self):
        epi = []
        for t in range(12):
            epi += self.epi_items(t)
        for _, fn in epi:
            fn()

    def kernel(self):
        p = self.p
        self.vm_log, self.lg_log = [], []
        self.prologue()
        tile_loop, body_plain, tile_done, no_prev, after_epi, finish = (p.newlabel(n) for n in ("tile", "plain", "tdone", "noprev", "aepi", "fin"))
        p.label(tile_loop)
        p.s_mov_b32(KBL, NKB)
        if self.deferred:
            # first body of a tile: C = 0 start + the deferred epilogue of the previous tile (skipped for the first tile)
            p.s_cmp_eq_u32(HAVEP, 0)
            p.s_cbranch_scc1(no_prev)
            self.body(True, True)
            p.s_branch(after_epi)
            p.label(no_prev)
            self.body(True, False)
            p.label(after_epi)
        else:
            self.body(True, False)
        p.s_sub_u32(KBL, KBL, 1)
        p.s_cmp_eq_u32(KBL, 0)
        p.s_cbranch_scc1(tile_done)
        p.label(body_plain)
        self.body(False, False)
        p.s_sub_u32(KBL, KBL, 1)
        p.s_cmp_lg_u32(KBL, 0)
        p.s_cbranch_scc1(body_plain)
        p.label(tile_done)
        self.tile_end()
        # the finished tile's identity for its epilogue; then advance the tile bookkeeping
        p.s_mov_b32(PO, CUR[2])
        p.s_mov_b32(PB, CB)
        p.s_add_u32(T0, PB, BIAS_LDS)
        p.v_add_u32(VB, T0, VBL)
        p.s_mov_b32(HAVEP, 1)
        if not self.deferred:
            self.exposed_epilogue()
        p.s_sub_u32(TLEFT, TLEFT, 1)
        p.s_cmp_eq_u32(TLEFT, 0)
        p.s_cbranch_scc1(finish)
        for k in range(4):
            p.s_mov_b32(CUR[k], NXT[k])
        p.s_add_u32(CIDX, CIDX, G16)
        self.load_next()
        # the new tile's bias into the other slot (its readers come a whole tile later, many barriers away)
        p.s_sub_u32(CB, 768, CB)
        self.bias_dma()
        p.s_waitcnt(lgkmcnt=0)
        p.s_branch(tile_loop)
        p.label(finish)
        if self.deferred:
            self.exposed_epilogue()
        p.s_waitcnt(vmcnt=0, lgkmcnt=0)
        return p


# ---------------------------------------------------------------------------------------------- host-side helpers
def tile_table(M, N, lda, ldw, ldc, grid=None, GR=4):
    """XCD-aware persistent tile order (the order gemm_q192 uses): virtual block vb runs on XCD vb % 8; every XCD owns a
    contiguous run of the tile order; tiles are ordered in groups of GR tile rows with the column index outer.
    Returns int32 [ntiles, 4] = {x_off, w_off, out_off, bias_off} in bytes, indexed by vb (the workgroup reads
    entries blockIdx.x, blockIdx.x + grid, ...)."""
    tiles_m, tiles_n = M // 256, N // 192
    ntiles = tiles_m * tiles_n
    tab = np.zeros((ntiles, 4), np.int32)
    q, r = ntiles >> 3, ntiles & 7
    for vb in range(ntiles):
        xcd, loc = vb & 7, vb >> 3
        t = (xcd * (q + 1) if xcd < r else r * (q + 1) + (xcd - r) * q) + loc
        group = t // (GR * tiles_n)
        within = t - group * GR * tiles_n
        first_m = group * GR
        gsz = min(GR, tiles_m - first_m)
        m0 = (first_m + within % gsz) * 256
        n0 = (within // gsz) * 192
        tab[vb] = (m0 * lda * 2, n0 * ldw * 2, (m0 * ldc + n0) * 2, n0 * 4)
    return tab


def make_kargs(A_, W_, bias, out, table, M, N, K, grid):
    """The 96-byte kernarg block (struct ZParams) as a numpy byte buffer + the pointer tags for the emulator."""
    ka = np.zeros(KARG_BYTES, np.uint8)
    dims = np.array([K * 2, K * 2, N * 2, table.shape[0], K // 768, grid, A_.size, W_.size, out.size, bias.size, 0, 0], np.int32)
    ka[K_DIMS:K_DIMS + 48] = dims.view(np.uint8)
    ptrs = {K_A: A_, K_W: W_, K_BIAS: bias, K_OUT: out, K_TABLE: table}
    return ka, ptrs


def emulate(prog, A16, W16, bias, act, grid, modes=(("eager", "eager", "0123"),), verbose=False):
    """Run the generated kernel on numpy operands for every mode tuple (dma, ds, wave order); returns the fp16 outputs."""
    M, K = A16.shape
    N = W16.shape[0]
    table = tile_table(M, N, K, K, N)
    ntiles = table.shape[0]
    grid = min(grid, ntiles)
    outs = []
    for dma, ds, order in modes:
        a_b = A16.view(np.uint8).reshape(-1).copy()
        w_b = W16.view(np.uint8).reshape(-1).copy()
        b_b = bias.astype(np.float32).view(np.uint8).reshape(-1).copy()
        o_b = np.full(M * N * 2, 0xFF, np.uint8)
        t_b = table.view(np.uint8).reshape(-1).copy()
        ka, ptrs = make_kargs(a_b, w_b, b_b, o_b, table, M, N, K, grid)
        for bid in range(grid):
            wg = Workgroup(prog, 4, 160 * 1024, dma_lazy=(dma == "lazy"), ds_lazy=(ds == "lazy"), order=order)
            wg.mem_objs[(id(ka), K_A)] = {0: a_b}
            wg.mem_objs[(id(ka), K_W)] = {0: w_b}
            wg.mem_objs[(id(ka), K_BIAS)] = {0: b_b}
            wg.mem_objs[(id(ka), K_OUT)] = {0: o_b}
            wg.mem_objs[(id(ka), K_TABLE)] = {0: t_b}
            for w, st in enumerate(wg.waves):
                st.s[KARG.idx] = 0x1000
                st.s[KARG.idx + 1] = 0
                st.sobj[KARG.idx] = ka
                st.s[BID.idx] = bid
                st.v[TID.idx] = (np.arange(64) + 64 * w).astype(np.uint32)
            wg.run()
            if verbose:
                print(f"  wg {bid}: {wg.executed} instructions, max VMEM in flight {max(s_.max_vm for s_ in wg.waves)}")
        outs.append(o_b.view(np.float16).reshape(M, N).copy())
    return outs


def reference(A16, W16, bias, act):
    x = A16.astype(np.float32) @ W16.astype(np.float32).T + bias.astype(np.float32)[None, :]
    if act == 1:
        from math import erf
        x = 0.5 * x * (1.0 + np.vectorize(erf)(x * 0.70710678118654752440))
    elif act == 2:
        x = np.maximum(x, 0)
    return x


CLOBBERS = ([f"v{i}" for i in range(4, 256)] + [f"a{i}" for i in range(256)] + [f"s{i}" for i in range(16, 80)] +
            ["vcc", "scc", "memory"])


def write_inc(path, prog):
    lines = prog.text().split("\n")
    with open(path, "w") as f:
        f.write("// GENERATED by tools/kgen/gemm_z192_gen.py — do not edit (tests/test_kgen_emulator.py checks it is current).\n")
        for ln in lines:
            f.write('"' + ln.replace("\\", "\\\\").replace('"', '\\"') + '\\n"\n')


def main():
    root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    for act in (0, 1):
        g = ZGen(act=act, deferred=True)
        prog = g.kernel()
        hz = check_hazards(prog)
        for h in hz[:20]:
            print("HAZARD:", h)
        assert not hz, f"{len(hz)} hazards"
        path = os.path.join(root, "sam_road_amd", "csrc", f"gemm_z192_body_act{act}.inc")
        write_inc(path, prog)
        print(f"act {act}: {prog.n_real()} instructions -> {path}")


if __name__ == "__main__":
    main()
