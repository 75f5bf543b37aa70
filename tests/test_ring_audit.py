"""k_score_wave hides its ring of block loads from the compiler (inline asm, hand-counted s_waitcnt vmcnt: the compiler's own
bookkeeping drains the ring at every step).  That is only sound while (cdna_hip_programming.md §5.7)
  * nothing else issues vector-memory instructions between the ring's first and last load (scratch spills included), and
  * the compiler does not copy a ring register to satisfy a wait statement's operand (a copy made before the data lands).
This test compiles the kernel to gfx950 assembly (no GPU needed) and checks both on the generated code."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "spaghettisearch_amd", "csrc", "score_wave.hip")


def test_ring_loads_are_alone_and_never_copied(tmp_path):
    out = tmp_path / "score_wave.s"
    subprocess.run(["/opt/rocm/bin/hipcc", "-std=c++17", "-O3", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math",
                    "-I/opt/rocm/include", "-S", "--cuda-device-only", "-o", str(out), SRC], check=True, capture_output=True)
    text = out.read_text().split("\n")
    start = next(i for i, l in enumerate(text) if l.startswith("_ZN3ssw12k_score_wave"))
    end = next(i for i in range(start, len(text)) if "s_endpgm" in text[i])
    body = text[start:end]
    loads = [i for i, l in enumerate(body) if "global_load_dwordx2" in l and "ASMSTART" in body[i - 1]]
    assert len(loads) >= 32, "the ring's asm loads are gone: is SSW_PLAIN_RING defined?"
    lo, hi = loads[0], loads[-1]
    ring = set()
    for i in loads:
        m = re.search(r"global_load_dwordx2 v\[(\d+):(\d+)\]", body[i])
        ring.update((int(m.group(1)), int(m.group(2))))
    for i in range(lo, hi):
        ins = body[i].strip()
        if not ins or ins.startswith((";", ".")):
            continue
        if re.match(r"(buffer_|flat_|global_)", ins):
            assert "ASMSTART" in body[i - 1], f"vector-memory instruction inside the ring region: {ins}"
        # (scratch reloads inside the region only make the hand-counted waits conservative: slower, never wrong)
        # a 64-bit copy of a ring register pair right in front of an asm wait = an operand copy made before the data landed
        m = re.match(r"v_mov_b64_e32 v\[\d+:\d+\], v\[(\d+):(\d+)\]", ins)
        if m and int(m.group(1)) in ring:
            nxt = [body[j].strip() for j in range(i + 1, min(i + 6, hi))]
            assert not any(x.startswith("s_waitcnt vmcnt") for x in nxt), f"ring register copied in front of a wait: {ins}"
