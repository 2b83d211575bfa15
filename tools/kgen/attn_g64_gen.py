#!/usr/bin/env python
"""EXPERIMENT (round 4, not shipped: parity with the HIP kernel, profiles/r04_attention_asm_global.txt).  Generator of a hand-scheduled
GLOBAL attention kernel (32 x 32 tokens, head dim 64: blocks 2 / 5 / 8 / 11 of ViT-B / ViT-L at 512 px) -> tools/probes/build/
attn_g64_body.inc, built into tools/probes/attn_win_probe only.  Same arithmetic, operation for operation, as
attn_global_kernel<32> in csrc/attention.hip (softmax(scale q.k + rel_h + rel_w) v with the decomposed
rel-pos bias from the unscaled q).  What changes is the schedule.

Why: the compiler's key loop leaves the SIMD idle a third of the time (PMC, profiles/r04_pmc_attn.txt: VALU busy 44 %, MFMA busy
24 %, never both) — a wave issues in order, so its S^T MFMAs, its softmax VALU block and its P.V MFMAs serialise, and two or three
waves per SIMD only partly fill each other's holes.  Here ONE wave per SIMD (4 waves, 512 registers each) owns TWO 32-query tiles
A and B and software-pipelines them against each other, an MFMA every ~8 VALU instructions:

    block X(n):  VALU softmax of A on key stage n      beside  MFMA  S^T of B, stage n      then  P.V of B, stage n-1
    block Y(n):  VALU softmax of B on key stage n      beside  MFMA  S^T of A, stage n+1    then  P.V of A, stage n

(key stage = 2 key tiles = 64 keys; 16 stages).  Every MFMA result is consumed one block later, every K / V fragment is read from
LDS once for both query tiles.  Registers: S^T tiles, rel_w, K / V fragments, P in VGPRs (what the VALU touches), the O^T
accumulators and the query fragments in AGPRs; rescaling O^T (rare: only when a row maximum moves) goes through v_accvgpr_read /
write in a cold block.  K rows and row-major V rows arrive by LDS-DMA into a 4-slot ring (one barrier per stage, DMA three stages
ahead), V^T fragments are read with ds_read_b64_tr_b16 (layouts: attention.hip's windowed kernel).

Emulated (tools/kgen/asmdsl.py, adversarial DMA / ds_read / wave-order modes) against a float64 reference by
tests/test_kgen_emulator.py.  Test infrastructure / build tooling: not part of the product path.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from asmdsl import A, Lit, M0, Prog, S, V, VCC, Workgroup, check_hazards, neg  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

# ---------------------------------------------------------------------------------------------- kernarg block (struct G64Params)
K_QKV, K_OUT, K_TH, K_TW = 0, 8, 16, 24
K_DIMS = 32          # ld2 (qkv row pitch, bytes), ldo2 (out row pitch, bytes), heads, nbh (images x heads)
K_F = 48             # scale, c_exp = scale * log2(e), 1 / scale, D2 = heads * 128 (bytes of one of q | k | v in a row)
K_DIV = 64           # magic, shift: bh / heads = (bh * magic) >> shift (host-checked exact for every bh < nbh); pad; pad
KARG_BYTES = 80

NKEY_STAGES = 16     # 1024 keys / 64
SLOT = 16384         # one ring slot: K tile a | K tile b | V tile a | V tile b
NSLOT = 4
RH_LDS = NSLOT * SLOT                 # rel_h tables: [wave][A | B][32 queries][34 f32] (k = 0..31, dump slot 32, pad)
RH_ROW = 34 * 4
RH_TILE = 32 * RH_ROW
LDS_BYTES = RH_LDS + 4 * 2 * RH_TILE  # 100 352 B

# ---------------------------------------------------------------------------------------------- registers
KARG, BID, WAVE = S(36, 2), S(38), S(39)
RS_Q, RS_O, RS_TH, RS_TW = S(40, 4), S(44, 4), S(48, 4), S(52, 4)
LD2, LDO2, HEADS, NBH = S(56), S(57), S(58), S(59)
SCALE, CEXP, INVS, D2 = S(60), S(61), S(62), S(63)
P_QKV, P_OUT, P_TH, P_TW = S(64, 2), S(66, 2), S(68, 2), S(70, 2)
HEAD, BIMG, QB = S(72), S(73), S(74)
T0, T1, T2, T3 = S(75), S(76), S(77), S(78)
MAGIC, SHIFT = S(90), S(91)
WLDS = S(81)             # wave * 1024: this wave's 8-row piece of every 32-row tile
SOFA, SOFB = S(82), S(83)    # DMA scalar offsets of the stage being fetched: tile a, tile b
LD64, LD32 = S(84), S(85)    # 64 * ld2, 32 * ld2
STG = S(86)              # next stage to fetch
ITER = S(87)
QSOFF = S(88)            # scalar offset of the q rows of tile B (32 * ld2)
OSOFF = S(89)            # 32 * ldo2
TH8 = S(92)              # 8.0f: the lazy-rescale threshold (not an inline constant)

TID, LANE = V(4), V(5)
SX = {"A": V(8, 32), "B": V(40, 32)}          # S^T of tiles a | b (16 + 16)
RELW = {"A": V(72, 16), "B": V(88, 16)}
KF = V(104, 32)          # K fragments: tile a ks 0..3, tile b ks 0..3 (4 registers each)
VF = V(136, 32)          # V^T fragments: tile a [dt][sx], tile b [dt][sx] (4 registers each)
PX = {"A": V(168, 16), "B": V(184, 16)}       # P as packed f16: tile a sx 0, 1, tile b sx 0, 1 (4 registers each)
MX = {"A": V(200), "B": V(202)}
LX = {"A": V(201), "B": V(203)}
RH = {0: {"A": V(204, 2), "B": V(206, 2)}, 1: {"A": V(208, 2), "B": V(210, 2)}}   # rel_h of the stage's two tiles, by stage parity
TMP = [V(212 + i) for i in range(16)]          # TMP[0:2], [2:4] ... even pairs
KAD = [V(228 + i) for i in range(4)]           # K fragment lane address per k-step (the swizzle is an XOR: not additive)
VAD = [V(232), V(233)]                         # V^T read lane bases, d tile 0 / 1
RHAD = {"A": V(234), "B": V(235)}              # row of the lane's query in the rel_h table (+ 8 bytes per stage)
KDMA, VDMA = V(236), V(237)                    # DMA source offsets of the lane (tile-relative)
OAD = V(238)
QAD = V(239)
TAD = V(240)
KB = [V(241), V(242)]                          # rel-pos scatter: lane base of k for table row tile jt
ROWB = V(243)
TF = V(104, 64)                                # prologue: rel-pos table fragments [table][jt][ks] in the K / V fragment registers
O = {"A": A(0, 32), "B": A(32, 32)}            # O^T accumulators: d tile 0 | 1
Q = {"A": A(64, 16), "B": A(80, 16)}           # query fragments, 4 k-steps

NEG_INF = 0xFF800000


def jr(r):
    return (r & 3) + 8 * (r >> 2)


class GGen:
    def __init__(self, sched=None):
        self.p = Prog()
        self.sched = dict(sched or {})
        self.cold = []                   # (label, back label, emitter) blocks laid out after s_endpgm-less tail

    # ------------------------------------------------------------------------------------------ prologue
    def prologue(self):
        p = self.p
        p.raw("s_mov_b64 s[36:37], %0")
        p.raw("s_mov_b32 s38, %1")
        p.raw("v_mov_b32 v4, %2")
        p.s_load_dwordx(2, P_QKV, KARG, K_QKV)
        p.s_load_dwordx(2, P_OUT, KARG, K_OUT)
        p.s_load_dwordx(2, P_TH, KARG, K_TH)
        p.s_load_dwordx(2, P_TW, KARG, K_TW)
        p.s_load_dwordx(4, S(56, 4), KARG, K_DIMS)
        p.s_load_dwordx(4, S(60, 4), KARG, K_F)
        p.s_load_dwordx(2, S(90, 2), KARG, K_DIV)
        p.v_and_b32(LANE, 63, TID)
        p.v_lshrrev_b32(TMP[0], 6, TID)
        p.s_nop(0)
        p.v_readfirstlane_b32(WAVE, TMP[0])
        p.s_waitcnt(lgkmcnt=0)
        # workgroup -> (image x head, 256-query block).  XCD-aware order (speed only): workgroup u runs on XCD u % 8; the four query
        # blocks of an (image, head) go to ONE XCD back to back so its K / V come from HBM once and from that L2 three more times.
        p.s_and_b32(T1, BID, 7)              # xcd
        p.s_lshr_b32(T2, BID, 3)             # j
        p.s_lshr_b32(T3, T2, 2)
        p.s_lshl_b32(T3, T3, 3)
        p.s_add_u32(T3, T3, T1)              # bh = (j / 4) * 8 + xcd
        p.s_and_b32(T2, T2, 3)               # qb = j % 4
        p.s_lshr_b32(T0, BID, 2)             # plain order: bh = u / 4, qb = u % 4
        p.s_and_b32(T1, BID, 3)
        p.s_and_b32(QB, NBH, 7)
        p.s_cmp_eq_u32(QB, 0)                # (the compare last: s_and / s_lshr / s_add all write SCC)
        p.s_cselect_b32(T3, T3, T0)
        p.s_cselect_b32(QB, T2, T1)
        # b = bh / heads by the host-checked multiply-shift, head = bh - b * heads
        p.s_mul_i32(T0, T3, MAGIC)
        p.s_lshr_b32(BIMG, T0, SHIFT)
        p.s_mul_i32(T0, BIMG, HEADS)
        p.s_sub_u32(HEAD, T3, T0)
        p.s_lshl_b32(WLDS, WAVE, 10)
        p.s_lshl_b32(LD64, LD2, 6)
        p.s_lshl_b32(LD32, LD2, 5)
        p.s_mov_b32(QSOFF, LD32)
        p.s_lshl_b32(OSOFF, LDO2, 5)
        p.s_mov_b32(TH8, 8.0)
        # buffer resources over this image's 1024 token rows
        p.s_lshl_b32(T0, LD2, 10)            # bytes per image of qkv
        p.s_mul_i32(T1, T0, BIMG)
        p.s_add_u32(RS_Q[0], P_QKV[0], T1)
        p.s_addc_u32(RS_Q[1], P_QKV[1], 0)
        p.s_and_b32(RS_Q[1], RS_Q[1], 0xFFFF)
        p.s_mov_b32(RS_Q[2], T0)
        p.s_mov_b32(RS_Q[3], 0x00020000)
        p.tag_copy(RS_Q, P_QKV)
        p.s_lshl_b32(T0, LDO2, 10)
        p.s_mul_i32(T1, T0, BIMG)
        p.s_add_u32(RS_O[0], P_OUT[0], T1)
        p.s_addc_u32(RS_O[1], P_OUT[1], 0)
        p.s_and_b32(RS_O[1], RS_O[1], 0xFFFF)
        p.s_mov_b32(RS_O[2], T0)
        p.s_mov_b32(RS_O[3], 0x00020000)
        p.tag_copy(RS_O, P_OUT)
        for rs, ptr in ((RS_TH, P_TH), (RS_TW, P_TW)):
            p.s_mov_b32(rs[0], ptr[0])
            p.s_and_b32(rs[1], ptr[1], 0xFFFF)
            p.s_mov_b32(rs[2], 63 * 128)
            p.s_mov_b32(rs[3], 0x00020000)
            p.tag_copy(rs, ptr)
        t0, t1, t2, t3 = TMP[0], TMP[1], TMP[2], TMP[3]
        # ---- q and rel-pos table fragment loads first (the oldest VMEM operations: they can be waited for with the DMAs in flight)
        # lane (q = lane & 31, half = lane >> 5): chunk (2 ks + half) of its query row -> Q[X][ks]
        p.v_and_b32(t0, 31, LANE)                               # q
        p.v_lshrrev_b32(t1, 5, LANE)                            # half
        p.s_lshl_b32(T0, QB, 8)
        p.s_lshl_b32(T1, WAVE, 6)
        p.s_add_u32(T0, T0, T1)                                 # first query of the wave (tile A); tile B = + 32
        p.v_add_u32(t2, T0, t0)                                 # qi of tile A
        p.v_mul_lo_u32(t3, t2, LD2)
        p.s_lshl_b32(T1, HEAD, 7)                               # head * 128 bytes
        p.v_lshl_add_u32(QAD, t1, 4, t3)
        p.v_add_u32(QAD, T1, QAD)                               # qi * ld2 + head * 128 + half * 16
        p.s_mov_b32(T2, 0)
        for X, so in (("A", T2), ("B", QSOFF)):
            for ks in range(4):
                p.buffer_load_dwordx4(Q[X].sub(4 * ks, 4), QAD, RS_Q, so, offset=32 * ks)
        # table fragment rows j = jt * 32 + (lane & 31), clamped to 62 (row 63 does not exist; its products land in the dump slot)
        for jt in range(2):
            p.v_add_u32(t2, 32 * jt, t0)
            p.v_min_u32(t2, 62, t2)
            p.v_lshlrev_b32(t2, 7, t2)
            p.v_lshl_add_u32(TAD, t1, 4, t2)                    # j * 128 + half * 16
            for ti, rs in enumerate((RS_TW, RS_TH)):
                for ks in range(4):
                    p.buffer_load_dwordx4(TF.sub(((ti * 2 + jt) * 4 + ks) * 4, 4), TAD, rs, T2, offset=32 * ks)
        # ---- DMA lane offsets: row r = wave * 8 + lane / 8 of a 32-row tile; K chunk (lane & 7) ^ ((r >> 1) & 7), V chunk (lane & 7) ^ 4 ((r >> 1) & 1)
        p.v_lshrrev_b32(t2, 3, LANE)
        p.v_lshl_add_u32(t2, WAVE, 3, t2)                       # r
        p.v_mul_lo_u32(t3, t2, LD2)
        p.v_add_u32(t3, T1, t3)                                 # r * ld2 + head * 128
        p.v_add_u32(t3, D2, t3)                                 # + D2: the k third of the row
        p.v_lshrrev_b32(TMP[4], 1, t2)
        p.v_and_b32(TMP[5], 7, TMP[4])
        p.v_and_b32(TMP[6], 7, LANE)
        p.v_xor_b32(TMP[5], TMP[5], TMP[6])
        p.v_lshl_add_u32(KDMA, TMP[5], 4, t3)
        p.v_and_b32(TMP[5], 1, TMP[4])
        p.v_lshlrev_b32(TMP[5], 2, TMP[5])
        p.v_xor_b32(TMP[5], TMP[5], TMP[6])
        p.v_add_u32(t3, D2, t3)                                 # the v third
        p.v_lshl_add_u32(VDMA, TMP[5], 4, t3)
        # ---- kick off stages 0, 1, 2
        p.s_mov_b32(STG, 0)
        for st in range(3):
            self.dma_stage(slot=st, fillers=None)
        # ---- fragment read addresses
        # K: row = lane & 31, chunk c = 2 ks + half at position c ^ ((row >> 1) & 7)
        p.v_lshrrev_b32(t2, 1, t0)
        p.v_and_b32(t2, 7, t2)                                  # (row >> 1) & 7
        p.v_lshlrev_b32(t3, 7, t0)                              # row * 128
        for ks in range(4):
            p.v_add_u32(TMP[4], 2 * ks, t1)
            p.v_xor_b32(TMP[4], TMP[4], t2)
            p.v_lshl_add_u32(KAD[ks], TMP[4], 4, t3)
        # V^T (ds_read_b64_tr_b16): i = lane & 15, g = (lane >> 4) & 1: (4 half + i / 4) * 128 + ((2 g + ((i >> 1) & 1)) ^ 4 ((i >> 3) & 1)) * 16 + (i & 1) * 8
        p.v_and_b32(TMP[4], 15, LANE)                           # i
        p.v_lshrrev_b32(TMP[5], 2, TMP[4])
        p.v_lshl_add_u32(TMP[5], t1, 2, TMP[5])                 # 4 half + i / 4
        p.v_lshlrev_b32(TMP[5], 7, TMP[5])
        p.v_lshrrev_b32(TMP[6], 4, LANE)
        p.v_and_b32(TMP[6], 1, TMP[6])                          # g
        p.v_lshrrev_b32(TMP[7], 1, TMP[4])
        p.v_and_b32(TMP[7], 1, TMP[7])
        p.v_lshl_add_u32(TMP[6], TMP[6], 1, TMP[7])             # 2 g + ((i >> 1) & 1)
        p.v_lshrrev_b32(TMP[7], 3, TMP[4])
        p.v_and_b32(TMP[7], 1, TMP[7])
        p.v_lshlrev_b32(TMP[7], 2, TMP[7])
        p.v_xor_b32(TMP[6], TMP[6], TMP[7])
        p.v_lshl_add_u32(TMP[5], TMP[6], 4, TMP[5])
        p.v_and_b32(TMP[7], 1, TMP[4])
        p.v_lshl_add_u32(VAD[0], TMP[7], 3, TMP[5])
        p.v_xor_b32(VAD[1], 64, VAD[0])
        # rel_h / scratch table rows of the lane's queries: RH_LDS + (wave * 2 + X) * RH_TILE + q * RH_ROW
        p.v_mul_lo_u32(t2, t0, self.const(RH_ROW, T3))
        p.s_mul_i32(T0, WAVE, 2 * RH_TILE)
        p.s_add_u32(T0, T0, RH_LDS)
        p.v_add_u32(RHAD["A"], T0, t2)
        p.s_add_u32(T0, T0, RH_TILE)
        p.v_add_u32(RHAD["B"], T0, t2)
        # output address: row (qi of tile A) * ldo2 + head * 128 + half * 16
        p.s_lshl_b32(T0, QB, 8)
        p.s_lshl_b32(T2, WAVE, 6)
        p.s_add_u32(T0, T0, T2)
        p.v_add_u32(t2, T0, t0)
        p.v_mul_lo_u32(t3, t2, LDO2)
        p.v_add_u32(t3, T1, t3)
        p.v_lshl_add_u32(OAD, t1, 4, t3)
        # ---- rel-pos bias of both query tiles
        p.s_waitcnt(vmcnt=12)                                   # everything older than the 12 DMAs: q and table fragments
        for X in ("A", "B"):
            self.relpos(X, t0, t1)
        # ---- state
        for X in ("A", "B"):
            p.v_mov_b32(MX[X], Lit(NEG_INF))
            p.v_mov_b32(LX[X], 0)
            for r in range(32):
                p.v_accvgpr_write_b32(O[X][r], 0)
        for r in range(16):
            p.v_mov_b32(PX["B"][r], 0)                          # block X(0) runs P.V of "stage -1" of B: 0 x 0
        for r in range(32):
            p.v_mov_b32(VF[r], 0)

    def const(self, val, sreg):
        """VOP3 takes no literal on gfx9: constants outside the inline range go through an SGPR"""
        self.p.s_mov_b32(sreg, val)
        return sreg

    def relpos(self, X, vq, vhalf):
        """rel_w -> RELW[X], rel_h -> the LDS table of (wave, X).  P^T[j, q] = T[j, :] . Q[q, :] for the 63 table rows (two 32-row MFMA
        tiles), scattered to rel[q][k = qc - j + 31] (min_u32 sends k < 0 and k > 31 to the dump slot 32)."""
        p = self.p
        acc = [V(8, 16), V(24, 16)]
        # qx = qi & 31 = lane & 31 (+ 0: the tile starts on a multiple of 32), qy = qi >> 5 = (qb * 256 + wave * 64 + X * 32) >> 5
        p.s_lshl_b32(T0, QB, 3)
        p.s_lshl_b32(T2, WAVE, 1)
        p.s_add_u32(T0, T0, T2)
        if X == "B":
            p.s_add_u32(T0, T0, 1)                              # qy (wave-uniform)
        for ti in range(2):                                     # 0: w table (qc = qx), 1: h table (qc = qy)
            for jt in range(2):
                for ks in range(4):
                    p.v_mfma_f32_32x32x16_f16(acc[jt], TF.sub(((ti * 2 + jt) * 4 + ks) * 4, 4), Q[X].sub(4 * ks, 4), 0 if ks == 0 else acc[jt])
            # lane base of k: qc + 31 - 4 half - 32 jt
            if ti == 0:
                p.v_lshlrev_b32(TMP[4], 2, vhalf)
                p.v_sub_u32(TMP[4], vq, TMP[4])                 # qx - 4 half
            else:
                p.v_lshlrev_b32(TMP[4], 2, vhalf)
                p.v_sub_u32(TMP[4], T0, TMP[4])                 # qy - 4 half
            p.v_add_u32(KB[0], 31, TMP[4])
            p.v_add_u32(KB[1], -1, TMP[4])
            p.s_nop(7)
            p.s_nop(3)
            for jt in range(2):
                for r in range(16):
                    t = TMP[6 + (r & 1) * 2]
                    p.v_sub_u32(t, KB[jt], jr(r))
                    p.v_min_u32(t, 32, t)
                    p.v_lshl_add_u32(t, t, 2, RHAD[X])
                    p.v_mul_f32(TMP[7 + (r & 1) * 2], INVS, acc[jt][r])
                    p.ds_write_b32(t, TMP[7 + (r & 1) * 2])
            if ti == 0:
                # the lane's 16 keys of a tile are window columns jr(r) + 4 half: two ds_read_b64 per group of four
                p.v_lshl_add_u32(TMP[4], vhalf, 4, RHAD[X])
                for m in range(4):
                    p.ds_read_b64(RELW[X].sub(4 * m, 2), TMP[4], offset=32 * m)
                    p.ds_read_b64(RELW[X].sub(4 * m + 2, 2), TMP[4], offset=32 * m + 8)
                p.s_waitcnt(lgkmcnt=0)

    # ------------------------------------------------------------------------------------------ pieces of a stage
    def dma_stage(self, slot, fillers):
        """The wave's four LDS-DMA pieces of stage STG into ring slot `slot`, then STG = min(STG + 1, 15) (the tail re-fetches the
        last stage: uniform vmcnt bookkeeping, no branch).  fillers: list to append single-instruction emitters to, or None = emit now."""
        p = self.p
        ops = []
        ops.append(lambda: p.s_mul_i32(SOFA, STG, LD64))
        ops.append(lambda: p.s_add_u32(SOFB, SOFA, LD32))
        for i, (vo, so) in enumerate(((KDMA, SOFA), (KDMA, SOFB), (VDMA, SOFA), (VDMA, SOFB))):
            ops.append(lambda i=i: p.s_add_u32(M0, WLDS, slot * SLOT + i * 4096))
            ops.append(lambda: p.s_nop(0))
            ops.append(lambda vo=vo, so=so: p.buffer_load_lds(16, vo, RS_Q, so))
        ops.append(lambda: p.s_add_u32(STG, STG, 1))
        ops.append(lambda: p.s_min_u32(STG, STG, NKEY_STAGES - 1))
        if fillers is None:
            for o in ops:
                o()
        else:
            fillers.extend(ops)

    def k_reads(self, slot):
        p = self.p
        return [lambda t=t, ks=ks: p.ds_read_b128(KF.sub((t * 4 + ks) * 4, 4), KAD[ks], offset=slot * SLOT + t * 4096)
                for t in range(2) for ks in range(4)]

    def v_reads(self, slot):
        p = self.p
        out = []
        for t in range(2):
            for dt in range(2):
                for sx in range(2):
                    base = ((t * 2 + dt) * 2 + sx) * 4
                    off = slot * SLOT + 8192 + t * 4096 + sx * 2048
                    out.append(lambda base=base, off=off, dt=dt: p.ds_read_b64_tr_b16(VF.sub(base, 2), VAD[dt], offset=off))
                    out.append(lambda base=base, off=off, dt=dt: p.ds_read_b64_tr_b16(VF.sub(base + 2, 2), VAD[dt], offset=off + 1024))
        return out

    def rh_reads(self, par, u):
        """rel_h of stage n + 1 = 4 it + u + 1 for both query tiles (RHAD advances 32 bytes per loop iteration)"""
        p = self.p
        return [lambda X=X: p.ds_read_b64(RH[par][X], RHAD[X], offset=8 * (u + 1)) for X in ("A", "B")]

    def s_mfmas(self, X):
        """S^T of query tile X on the K fragments in KF: tiles a, b alternating, 4 k-steps (rel_w is the first C operand)"""
        p = self.p
        out = []
        for ks in range(4):
            for t in range(2):
                d = SX[X].sub(16 * t, 16)
                if self.sched.get("s_in_agpr"):       # timing experiment (wrong results): S^T accumulators in AGPRs — does the MFMA then run beside the VALU?
                    d = A(96 + (32 if X == "B" else 0) + 16 * t, 16)
                    out.append(lambda d=d, t=t, ks=ks: p.v_mfma_f32_32x32x16_f16(d, KF.sub((t * 4 + ks) * 4, 4), Q[X].sub(4 * ks, 4), d))
                    continue
                out.append(lambda d=d, t=t, ks=ks: p.v_mfma_f32_32x32x16_f16(d, KF.sub((t * 4 + ks) * 4, 4), Q[X].sub(4 * ks, 4), RELW[X] if ks == 0 else d))
        return out

    def pv_mfmas(self, X):
        p = self.p
        out = []
        for t in range(2):
            for sx in range(2):
                for dt in range(2):
                    d = O[X].sub(16 * dt, 16)
                    a = VF.sub(((t * 2 + dt) * 2 + sx) * 4, 4)
                    b = PX[X].sub((t * 2 + sx) * 4, 4)
                    out.append(lambda d=d, a=a, b=b: p.v_mfma_f32_32x32x16_f16(d, a, b, d))
        return out

    def softmax(self, X, par, tag):
        """The VALU stream of one query tile and key stage (124 instructions), operation for operation attn_tile2<32>."""
        p = self.p
        sa, sb = SX[X].sub(0, 16), SX[X].sub(16, 16)
        rh = RH[par][X]
        t0, t1, mnew, mc, mca, mcb = TMP[0], TMP[1], TMP[2], TMP[3], TMP[4], TMP[5]
        sum2 = V(TMP[6].idx, 2)
        ops = []
        e = ops.append
        for s_, t in ((sa, t0), (sb, t1)):
            e(lambda s_=s_, t=t: p.v_max3_f32(t, s_[0], s_[1], s_[2]))
            for r in range(3, 15, 2):
                e(lambda s_=s_, t=t, r=r: p.v_max3_f32(t, t, s_[r], s_[r + 1]))
            e(lambda s_=s_, t=t: p.v_max_f32(t, t, s_[15]))
        e(lambda: p.v_add_f32(t0, t0, rh[0]))
        e(lambda: p.v_add_f32(t1, t1, rh[1]))
        e(lambda: p.v_max_f32(t0, t0, t1))
        e(lambda: p.v_mov_b32(t1, t0))
        e(lambda: p.s_nop(1))
        e(lambda: p.v_permlane32_swap_b32(t0, t1))               # t0 = {lo, lo}, t1 = {hi, hi}
        e(lambda: p.v_max3_f32(mnew, MX[X], t0, t1))
        # sched lazy: MX is the REFERENCE of the exponentials and moves only when some row's maximum has outgrown it by more than 2^8
        # (P up to 2^8); 6 500 fewer instructions per workgroup, no measurable time
        if self.sched.get("lazy"):
            e(lambda: p.v_sub_f32(t0, mnew, MX[X]))
            e(lambda: p.v_mul_f32(t0, CEXP, t0))
            e(lambda: p.v_cmp_lt_f32(TH8, t0))
        else:                                                    # attention.hip's rule: whenever some row maximum moved
            e(lambda: p.v_cmp_neq_f32(mnew, MX[X]))
        lab, back = p.newlabel(f"rescale_{X}_{tag}"), p.newlabel(f"back_{X}_{tag}")

        def branch():
            p.s_cbranch_vccnz(lab)
            p.label(back)
        e(branch)
        self.cold.append((lab, back, X, mnew))
        e(lambda: p.v_mul_f32(mc, neg(MX[X]), CEXP))
        e(lambda: p.v_fma_f32(mca, rh[0], CEXP, mc))
        e(lambda: p.v_fma_f32(mcb, rh[1], CEXP, mc))
        for s_, m_ in ((sa, mca), (sb, mcb)):
            for r in range(16):
                e(lambda s_=s_, m_=m_, r=r: p.v_fma_f32(s_[r], s_[r], CEXP, m_))
                if r & 1:
                    e(lambda s_=s_, r=r: p.v_exp_f32(s_[r - 1], s_[r - 1]))
                    e(lambda s_=s_, r=r: p.v_exp_f32(s_[r], s_[r]))
        first = True
        for t, s_ in enumerate((sa, sb)):
            for r in range(0, 16, 2):
                pr = s_.sub(r, 2)
                # (two plain adds per pair: a v_pk_add_f32 costs ~13 cycles more than its slot beside MFMAs, MI355X guide)
                if first:
                    first = False
                    nxt = s_.sub(2, 2)
                    e(lambda pr=pr, nxt=nxt: p.v_add_f32(sum2[0], pr[0], nxt[0]))
                    e(lambda pr=pr, nxt=nxt: p.v_add_f32(sum2[1], pr[1], nxt[1]))
                elif not (t == 0 and r == 2):
                    e(lambda pr=pr: p.v_add_f32(sum2[0], sum2[0], pr[0]))
                    e(lambda pr=pr: p.v_add_f32(sum2[1], sum2[1], pr[1]))
            for r in range(0, 16, 2):
                d = PX[X][(t * 2 + (r >> 3)) * 4 + ((r & 7) >> 1)]
                e(lambda d=d, s_=s_, r=r: p.v_cvt_pk_f16_f32(d, s_[r], s_[r + 1]))
        e(lambda: p.v_add_f32(t0, sum2[0], sum2[1]))
        e(lambda: p.v_add_f32(LX[X], LX[X], t0))
        return ops

    def emit_cold(self):
        """Rescale blocks: O^T lives in AGPRs, so alpha goes through v_accvgpr_read / v_mul / v_accvgpr_write (rare: only when some row
        maximum of the tile moved)."""
        p = self.p
        for lab, back, X, mnew in self.cold:
            p.label(lab)
            al = TMP[8]
            p.s_nop(15)                                          # the tile's last P.V MFMAs may still be in flight
            p.v_sub_f32(al, MX[X], mnew)
            p.v_mul_f32(al, CEXP, al)
            p.v_exp_f32(al, al)
            p.v_mov_b32(MX[X], mnew)
            p.s_nop(0)
            p.v_mul_f32(LX[X], al, LX[X])
            for r in range(0, 32, 2):
                pr = V(TMP[10].idx, 2)
                p.v_accvgpr_read_b32(pr[0], O[X][r])
                p.v_accvgpr_read_b32(pr[1], O[X][r + 1])
                p.v_mul_f32(pr[0], al, pr[0])
                p.v_mul_f32(pr[1], al, pr[1])
                p.v_accvgpr_write_b32(O[X][r], pr[0])
                p.v_accvgpr_write_b32(O[X][r + 1], pr[1])
            p.s_branch(back)
        self.cold = []

    def interleave(self, mfmas, fillers, lead=0):
        """One MFMA, then an even share of the fillers, ...; `lead` fillers go first (before the first MFMA)."""
        p = self.p
        fillers = list(fillers)
        for _ in range(min(lead, len(fillers))):
            fillers.pop(0)()
        n = len(mfmas)
        for i, m in enumerate(mfmas):
            m()
            k = (len(fillers) + (n - i) - 1) // (n - i) if n - i else len(fillers)
            for _ in range(k):
                fillers.pop(0)()
        for f in fillers:
            f()

    # ------------------------------------------------------------------------------------------ the loop
    def stage(self, u):
        """Key stage n = 4 it + u: blocks X(n) and Y(n).  Ring slots: stage n in slot u, n + 1 in slot (u + 1) % 4, the DMA fetches
        stage n + 3 into slot (u + 3) % 4 (= stage n - 1's, fully read before this stage's barrier)."""
        p = self.p
        par = u & 1
        # ---- X(n)
        p.s_waitcnt(vmcnt=4)                                     # this wave's pieces of stage n + 1 have landed (stage n + 2's may be in flight)
        if not self.sched.get("no_barrier"):
            p.s_barrier()                                        # ... every wave's; and every wave is done with stage n - 1's slot
        ab = self.sched
        keep = lambda name, ops: [] if ab.get("no_" + name) else ops          # probe variants (ablations: wrong results)
        fill = []
        self.dma_stage(slot=(u + 3) % 4, fillers=fill)
        fill = keep("dma", fill)
        sm = [] if ab.get("no_softmax") else self.softmax("A", par, f"x{u}")
        half = len(sm) // 2
        if ab.get("mix_mfma"):          # timing experiment: the 16 MFMAs of the block round-robin over the four accumulator chains
            sm_, pv_ = self.s_mfmas("B"), self.pv_mfmas("B")
            mixed = [m for pr in zip(sm_, pv_) for m in pr]
            self.interleave(keep("mfma", mixed), fill + sm + keep("lds", self.k_reads((u + 1) % 4) + self.rh_reads(par ^ 1, u)))
        else:
            self.interleave(keep("mfma", self.s_mfmas("B")), fill + sm[:half])
            self.interleave(keep("mfma", self.pv_mfmas("B")), keep("lds", self.k_reads((u + 1) % 4) + self.rh_reads(par ^ 1, u)) + sm[half:])
        for f in keep("lds", self.v_reads(u)):
            f()
        # ---- Y(n)
        p.s_waitcnt(lgkmcnt=15)                                  # 26 reads queued in order K, rel_h, V: the first 11 are back
        sm = [] if ab.get("no_softmax") else self.softmax("B", par, f"y{u}")
        half = len(sm) // 2
        if ab.get("mix_mfma"):
            p.s_waitcnt(lgkmcnt=0)
            sm_, pv_ = self.s_mfmas("A"), self.pv_mfmas("A")
            self.interleave(keep("mfma", [m for pr in zip(sm_, pv_) for m in pr]), sm)
        else:
            self.interleave(keep("mfma", self.s_mfmas("A")), sm[:half])
            p.s_waitcnt(lgkmcnt=0)                                   # V fragments of stage n
            self.interleave(keep("mfma", self.pv_mfmas("A")), sm[half:])

    def kernel(self):
        p = self.p
        self.prologue()
        # ---- stage 0's K fragments, rel_h, and S^T of A
        p.s_waitcnt(vmcnt=8)
        p.s_barrier()
        for f in self.k_reads(0):
            f()
        for X in ("A", "B"):
            p.ds_read_b64(RH[0][X], RHAD[X], offset=0)
        p.s_waitcnt(lgkmcnt=0)
        for m in self.s_mfmas("A"):
            m()
        # the loop's first barrier also waits for stage 1: three stages are in flight / landed, STG = 3
        p.s_mov_b32(ITER, 0)
        loop = p.newlabel("stage_loop")
        p.label(loop)
        for u in range(4):
            if not self.sched.get("no_loop"):
                self.stage(u)
        for X in ("A", "B"):
            p.v_add_u32(RHAD[X], 32, RHAD[X])
        p.s_add_u32(ITER, ITER, 1)
        p.s_cmp_lt_u32(ITER, NKEY_STAGES // 4)
        p.s_cbranch_scc1(loop)
        # ---- tail: P.V of B for the last stage beside the normalisation of A, then B
        self.interleave(self.pv_mfmas("B"), self.normalise("A"))
        self.store("A")
        p.s_nop(7)
        p.s_nop(7)
        for f in self.normalise("B"):
            f()
        self.store("B")
        p.s_waitcnt(vmcnt=0)
        done = p.newlabel("done")
        p.s_branch(done)
        self.emit_cold()
        p.label(done)
        return p

    # ------------------------------------------------------------------------------------------ epilogue
    def normalise(self, X):
        """O^T / row sum -> packed f16 in SX[X] (its 32 registers are free): register 2 * (dt * 4 + qd) + {0, 1} = dims 8 qd + 4 half + 0..3
        of d tile dt; then the halves trade 4-dim pieces so that every lane holds 8 consecutive dims (16-byte stores)."""
        p = self.p
        t0, t1, inv = TMP[0], TMP[1], TMP[2]
        pk = SX[X]
        ops = []
        e = ops.append
        e(lambda: p.v_mov_b32(t0, LX[X]))
        e(lambda: p.v_mov_b32(t1, LX[X]))
        e(lambda: p.s_nop(1))
        e(lambda: p.v_permlane32_swap_b32(t0, t1))
        e(lambda: p.v_add_f32(t0, t0, t1))
        e(lambda: p.v_rcp_f32(inv, t0))
        for dt in range(2):
            for r in range(0, 16, 2):
                a, b = V(TMP[4].idx + (r & 2) * 2), V(TMP[5].idx + (r & 2) * 2)
                e(lambda a=a, dt=dt, r=r: p.v_accvgpr_read_b32(a, O[X][dt * 16 + r]))
                e(lambda b=b, dt=dt, r=r: p.v_accvgpr_read_b32(b, O[X][dt * 16 + r + 1]))
                e(lambda a=a: p.v_mul_f32(a, a, inv))
                e(lambda b=b: p.v_mul_f32(b, b, inv))
                e(lambda a=a, b=b, dt=dt, r=r: p.v_cvt_pk_f16_f32(pk[dt * 8 + (r >> 1)], a, b))
        for dt in range(2):
            for j in range(2):
                for h in range(2):
                    lo, hi = pk[dt * 8 + 4 * j + h], pk[dt * 8 + 4 * j + 2 + h]
                    e(lambda: p.s_nop(0))
                    e(lambda lo=lo, hi=hi: p.v_permlane32_swap_b32(lo, hi))
        return ops

    def store(self, X):
        p = self.p
        p.s_mov_b32(T0, 0)
        so = T0 if X == "A" else OSOFF
        # lane base + d-block: half 0 lanes hold dims 16 j .. 16 j + 7 of d tile dt, half 1 lanes 16 j + 8 .. 16 j + 15 (OAD has + 16 half)
        for dt in range(2):
            for j in range(2):
                p.v_add_u32(TMP[8 + (dt * 2 + j) % 4], (dt * 32 + 16 * j) * 2, OAD)
        p.s_nop(0)
        for dt in range(2):
            for j in range(2):
                p.buffer_store_dwordx4(SX[X].sub(dt * 8 + 4 * j, 4), TMP[8 + (dt * 2 + j) % 4], RS_O, so)


# ---------------------------------------------------------------------------------------------- host side / emulation
def magic_div(heads, nbh):
    for shift in range(8, 24):
        magic = -(-(1 << shift) // heads)
        if all((bh * magic) >> shift == bh // heads for bh in range(max(nbh, 1))) and nbh * magic < (1 << 31):
            return magic, shift
    raise ValueError((heads, nbh))


def make_kargs(ld, ldo, heads, nbh, scale):
    ka = np.zeros(KARG_BYTES, np.uint8)
    ka[K_DIMS:K_DIMS + 16] = np.array([ld * 2, ldo * 2, heads, nbh], np.int32).view(np.uint8)
    c_exp = np.float32(np.float32(scale) * np.float32(1.4426950408889634))
    ka[K_F:K_F + 12] = np.array([scale, c_exp, np.float32(1.0) / np.float32(scale)], np.float32).view(np.uint8)
    ka[K_F + 12:K_F + 16] = np.array([heads * 128], np.int32).view(np.uint8)
    ka[K_DIV:K_DIV + 8] = np.array(magic_div(heads, nbh), np.int32).view(np.uint8)
    return ka


def emulate(prog, qkv16, th16, tw16, B, heads, scale=0.125, modes=(("eager", "eager", "0123"),), wgs=None, verbose=False):
    """qkv16 [B * 1024, 3 * heads * 64] f16; tables [63, 64] f16.  Returns out [B * 1024, heads * 64] f16 per mode (rows of workgroups
    not in `wgs` stay 0xFFFF)."""
    D = heads * 64
    nbh = B * heads
    outs = []
    for dma, ds, order in modes:
        q_b = qkv16.view(np.uint8).reshape(-1).copy()
        h_b = th16.view(np.uint8).reshape(-1).copy()
        w_b = tw16.view(np.uint8).reshape(-1).copy()
        o_b = np.full(B * 1024 * D * 2, 0xFF, np.uint8)
        ka = make_kargs(3 * D, D, heads, nbh, scale)
        for bid in (wgs if wgs is not None else range(nbh * 4)):
            wg = Workgroup(prog, 4, LDS_BYTES, dma_lazy=(dma == "lazy"), ds_lazy=(ds == "lazy"), order=order)
            wg.mem_objs[(id(ka), K_QKV)] = {0: q_b}
            wg.mem_objs[(id(ka), K_OUT)] = {0: o_b}
            wg.mem_objs[(id(ka), K_TH)] = {0: h_b}
            wg.mem_objs[(id(ka), K_TW)] = {0: w_b}
            for w, st in enumerate(wg.waves):
                st.s[KARG.idx], st.s[KARG.idx + 1] = 0x1000, 0
                st.sobj[KARG.idx] = ka
                st.s[BID.idx] = bid
                st.v[TID.idx] = (np.arange(64) + 64 * w).astype(np.uint32)
            wg.run()
            if verbose:
                print(f"  wg {bid}: {wg.executed} instructions, max VMEM in flight {max(s_.max_vm for s_ in wg.waves)}", flush=True)
        outs.append(o_b.view(np.float16).reshape(B * 1024, D).copy())
    return outs


def wg_rows(bid, B, heads):
    """(image, head, first query) of workgroup bid — the kernel's own mapping"""
    nbh = B * heads
    if nbh % 8 == 0:
        xcd, j = bid & 7, bid >> 3
        bh, qb = (j >> 2) * 8 + xcd, j & 3
    else:
        bh, qb = bid >> 2, bid & 3
    return bh // heads, bh % heads, qb * 256


def reference(qkv16, th16, tw16, B, heads, scale=0.125):
    """float64 attention with the decomposed rel-pos bias on the f16 inputs"""
    D = heads * 64
    x = qkv16.astype(np.float64).reshape(B, 1024, 3, heads, 64)
    q, k, v = x[:, :, 0], x[:, :, 1], x[:, :, 2]
    th, tw = th16.astype(np.float64), tw16.astype(np.float64)
    idx = np.arange(32)[:, None] - np.arange(32)[None, :] + 31          # [q coordinate, k coordinate] -> table row
    out = np.zeros((B, 1024, heads, 64))
    for b in range(B):
        for h in range(heads):
            s = scale * q[b, :, h] @ k[b, :, h].T                       # [1024 q, 1024 k]
            rh = np.einsum("qd,qkd->qk", q[b, :, h], th[idx][np.arange(1024) >> 5])     # [q, kh]
            rw = np.einsum("qd,qkd->qk", q[b, :, h], tw[idx][np.arange(1024) & 31])     # [q, kw]
            s = s + rh[:, np.arange(1024) >> 5] + rw[:, np.arange(1024) & 31]
            s = s - s.max(axis=1, keepdims=True)
            pr = np.exp(s)
            out[b, :, h] = (pr / pr.sum(axis=1, keepdims=True)) @ v[b, :, h]
    return out.reshape(B * 1024, D)


def write_inc(path, prog):
    with open(path, "w") as f:
        f.write("// GENERATED by tools/kgen/attn_g64_gen.py — do not edit (tests/test_kgen_emulator.py checks it is current).\n")
        for ln in prog.text().split("\n"):
            f.write('"' + ln.replace("\\", "\\\\").replace('"', '\\"') + '\\n"\n')


def write_meta(path):
    with open(path, "w") as f:
        f.write("// GENERATED by tools/kgen/attn_g64_gen.py — do not edit.\n")
        f.write(f"#define G64_LDS_BYTES {LDS_BYTES}      // K / V ring (4 x 16 KiB) + rel_h tables\n")
        f.write(f"#define G64_KARG_BYTES {KARG_BYTES}\n")


VARIANTS = {       # probe builds: tools/probes/attn_win_probe, ablate 21.. (ablations give wrong results)
    1: dict(no_softmax=True),                 # MFMA + LDS + DMA only
    2: dict(no_mfma=True),                    # VALU + LDS + DMA only
    3: dict(no_lds=True),
    4: dict(no_dma=True, no_barrier=True),
    5: dict(no_softmax=True, no_lds=True, no_dma=True, no_barrier=True),      # MFMAs alone
    6: dict(no_mfma=True, no_lds=True, no_dma=True, no_barrier=True),         # the softmax VALU stream alone
    7: dict(s_in_agpr=True),                                                  # S^T accumulators in AGPRs (results wrong): MFMA beside VALU?
    8: dict(no_loop=True),                                                    # prologue + epilogue only
}


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--variants":
        for v, sched in VARIANTS.items():
            prog = GGen(sched=sched).kernel()
            write_inc(os.path.join(sys.argv[2], f"attn_g64_var{v}.inc"), prog)
        return
    prog = GGen().kernel()
    probs = check_hazards(prog, verbose=True)
    assert not probs, f"{len(probs)} hazards"
    outdir = os.path.join(ROOT, "tools", "probes", "build")
    os.makedirs(outdir, exist_ok=True)
    out = os.path.join(outdir, "attn_g64_body.inc")
    write_inc(out, prog)
    write_meta(os.path.join(outdir, "attn_g64_meta.inc"))
    print(f"{prog.n_real()} instructions -> {out}")


if __name__ == "__main__":
    main()
