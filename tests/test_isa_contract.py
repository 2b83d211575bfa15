"""The LDS-DMA rings written in HIP (gemm_ring_kernel, attn_global_kernel) rely on a compiler behaviour, not on a documented
guarantee: with one static __shared__ array per ring stage and __builtin_amdgcn_s_waitcnt for the counted wait, hipcc adds NO
s_waitcnt vmcnt(0) of its own in front of the LDS reads of the other stages.  If a toolchain update changes that, the
kernels stay correct but silently lose their overlap — so the generated ISA is checked here: inside the ring loops the only vmcnt
waits are the ones the source asks for.  CPU only (hipcc -S cross-compiles for gfx950)."""
import os
import re
import shutil
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = next((c for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")) if c and os.path.exists(c)), None)
pytestmark = pytest.mark.skipif(HIPCC is None, reason="hipcc not found")


def _isa(src, extra=()):
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", *extra,
               os.path.join(ROOT, "sam_road_amd", "csrc", src), "-o", out]
        r = subprocess.run(cmd, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        return open(out).read()


def _kernel_body(isa, mangled_fragment):
    m = re.search(r"^(_Z\w*" + re.escape(mangled_fragment) + r"\w*):", isa, re.M)
    assert m, f"kernel {mangled_fragment} not found"
    a = m.start()
    return isa[a:isa.index(".Lfunc_end", a)]


def _events(body):
    ev = []
    for ln in body.split("\n"):
        t = ln.split(";")[0].strip()
        if t.startswith("s_barrier"):
            ev.append("B")
        elif re.match(r"s_waitcnt .*vmcnt\((\d+)\)", t):
            ev.append(int(re.search(r"vmcnt\((\d+)\)", t).group(1)))
        elif re.match(r"(buffer_load\w* .* lds|global_load_lds)", t):
            ev.append("D")
        elif t.startswith("ds_read"):
            ev.append("R")
    return ev


def test_gemm_ring_loop_has_only_the_counted_wait():
    body = _kernel_body(_isa("gemm.hip"), "gemm_ring_kernelILi3E")
    ev = _events(body)
    # the loop: [vmcnt(8), barrier, 8 DMA, fragment reads] x 3 per trip; find the three steps by their counted wait
    idx = [i for i, e in enumerate(ev) if e == 8]
    assert len(idx) >= 3, ev[:80]
    for i in idx[:3]:
        assert ev[i + 1] == "B", ev[i:i + 4]
        step = []
        for e in ev[i + 2:]:                        # the step's DMA issue and fragment reads: up to the last read before the next barrier / wait-then-barrier
            if e == "B" or (isinstance(e, int) and len([x for x in step if x == "R"]) >= 16):
                break
            step.append(e)
        assert step.count("D") == 8 and step.count("R") >= 16, step
        assert not [e for e in step if isinstance(e, int)], f"compiler-inserted vmcnt wait inside a ring step: {step}"


def test_global_attention_ring_has_one_wait_per_stage():
    body = _kernel_body(_isa("attention.hip", ["-fno-honor-nans"]), "attn_global_kernelILi32ELi3E")
    ev = _events(body)
    # after the prologue every stage is: vmcnt(0) (ours), barrier, then — in whatever order the compiler likes — the rel_h reads, the next
    # stage's 4 DMA pieces and the K / V fragment reads, and nothing else that waits: segments between barriers that carry 4 DMA pieces
    segs, cur = [], []
    for e in ev:
        if e == "B":
            segs.append(cur)
            cur = []
        else:
            cur.append(e)
    segs.append(cur)
    stages = [sg for sg in segs[1:] if sg.count("D") == 4]
    assert len(stages) >= 2, ev
    for sg in stages:
        assert sg.count("R") >= 8, sg
        waits = [x for x in sg if isinstance(x, int)]
        assert waits == [0] and sg[-1] == 0, f"a stage carries the waits {waits} (expected exactly its own hand-over vmcnt(0), at its end): {sg}"


def _vmem_events(body):
    """Every VMEM operation and vmcnt wait of a kernel body, in program order: 'D' LDS-DMA, 'L' load, 'S' store, int = vmcnt(n), 'B' barrier."""
    ev = []
    for ln in body.split("\n"):
        t = ln.split(";")[0].strip()
        if t.startswith("s_barrier"):
            ev.append("B")
        elif re.match(r"s_waitcnt .*vmcnt\((\d+)\)", t):
            ev.append(int(re.search(r"vmcnt\((\d+)\)", t).group(1)))
        elif re.match(r"(buffer_load\w* .* lds|global_load_lds)", t):
            ev.append("D")
        elif re.match(r"(global|buffer|flat|scratch)_load", t):
            ev.append("L")
        elif re.match(r"(global|buffer|flat|scratch)_store", t):
            ev.append("S")
    return ev


def test_attention_output_stores_are_16_bytes():
    """store_query: one v_permlane32_swap per pair of register quads, then 4 stores of 16 bytes per lane (was 8 of 8 bytes) — the store
    path's cost is per instruction (32 token rows each either way)."""
    body = _kernel_body(_isa("attention.hip", ["-fno-honor-nans"]), "attn_window_kernel")
    assert len(re.findall(r"global_store_dwordx4", body)) >= 4 and not re.findall(r"global_store_dwordx2", body)
    assert len(re.findall(r"v_permlane32_swap", body)) >= 8


def test_gemm_r320_loop_is_seven_dma_pieces_and_nothing_else():
    """gemm_r320_kernel (two stages, inline-asm LDS-DMA, inline-asm vmcnt(0) + barrier per k-tile): the wait is exact only if nothing but
    the 7 pieces of the next k-tile is issued between two waits — no compiler-inserted VMEM operation (a spill), no compiler-inserted wait."""
    body = _kernel_body(_isa("gemm.hip"), "gemm_r320_kernel")
    assert "scratch_" not in body
    ev = _vmem_events(body)
    i = ev.index(0)                                          # the loop's wait
    assert ev[:i] == ["D"] * 7, ev[:i]                       # prologue: k-tile 0
    assert ev[i + 1] == "B"
    j = i + 2
    trip = []
    while j < len(ev) and ev[j] != 0:
        trip.append(ev[j])
        j += 1
    assert trip == ["D"] * 7, trip


def test_gemm_pp_slots_hold_exactly_what_the_counted_waits_assume():
    """gemm_pp_kernel<0> (ping-pong k-loop, inline-asm LDS-DMA and counted waits): per k-tile a wave reads 16 fragments, sends exactly 6
    pieces, computes 16 MFMAs between two barriers — and the only vmcnt waits are the group's own (6: the next k-tile landed, the one
    after may fly; 0 on the last k-tile).  Anything else the compiler added (a spill, a hoisted load, a conservative wait) breaks the
    count or the overlap."""
    body = _kernel_body(_isa("gemm.hip"), "gemm_pp_kernelILi0E")
    assert "scratch_" not in body
    ev = []
    for ln in body.split("\n"):
        t = ln.split(";")[0].strip()
        if t.startswith("s_barrier"):
            ev.append("B")
        elif re.match(r"s_waitcnt .*vmcnt\((\d+)\)", t):
            ev.append(int(re.search(r"vmcnt\((\d+)\)", t).group(1)))
        elif "global_load_lds" in t:
            ev.append("D")
        elif re.match(r"(global|buffer|flat|scratch)_(load|store)", t):
            ev.append("V")
        elif t.startswith("ds_read"):
            ev.append("R")
        elif t.startswith("v_mfma"):
            ev.append("M")
    first_m = ev.index("M")
    last_m = len(ev) - 1 - ev[::-1].index("M")
    pro, loop = ev[:first_m], ev[first_m:last_m + 1]
    assert pro.count("D") == 18 and "V" not in pro, pro       # two k-tiles before the loop + the first memory slot's six
    assert pro.count("R") == 16, pro
    assert loop.count("M") == 16 and "V" not in loop and "D" not in loop and "R" not in loop, loop     # the compute slot is MFMAs only
    # the memory slot (between the loop's closing barrier and the barrier before the MFMAs): reads first, then six pieces, then the wait
    k = len(pro) - 1 - pro[::-1].index("B")                   # the barrier that opens the compute slot
    j = k - 1
    while pro[j] != "B":
        j -= 1
    slot = pro[j + 1:k]
    assert [e for e in slot if e in ("R", "D")] == ["R"] * 16 + ["D"] * 6, slot
    assert [e for e in slot if isinstance(e, int)] in ([6, 0], [0, 6]), slot          # group 1's two alternatives, nothing else
    tail = ev[last_m + 1:]                                    # group 1's closing barrier, then group 0's wait (6, or 0 on the last k-tile) and barrier
    assert tail[:4] in (["B", 6, 0, "B"], ["B", 0, 6, "B"], [6, 0, "B", "B"], [0, 6, "B", "B"]), tail[:8]
