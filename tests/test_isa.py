"""ISA-level regression guards for the token loop (no GPU: hipcc cross-compiles gfx950).  The kernels of the decode step are 5 - 10 us
long; what sits between kernel entry and the first memory request is on their critical path (NOTES.md).  These tests read the
cross-compiled assembly: the arguments the first requests need must arrive preloaded in SGPRs, and the fast projection kernels must
issue their weight requests before they wait for anything fetched from the argument segment."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "bioreason_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"


def _makefile_flags():
    mk = open(os.path.join(CSRC, "Makefile")).read()
    line = next(l for l in mk.splitlines() if l.startswith("HIPFLAGS :="))
    flags = line.split(":=", 1)[1].replace("$(ARCH)", "gfx950").replace("$(EXTRA_HIPFLAGS)", "").split()
    return [f for f in flags if f != "-I."]


def _asm(src, tmp_path):
    # `make isa` (run by __graft_entry__.build() beside the library) leaves the same assembly under csrc/build: taken when it is
    # newer than every source it depends on, otherwise compiled here (1-2 minutes for k_decgemm.hip)
    made = os.path.join(CSRC, "build", src.replace(".hip", ".s"))
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h") or f in (src, "Makefile")]      # = the Makefile rule's
    if os.path.exists(made) and os.path.getmtime(made) >= max(os.path.getmtime(d) for d in deps):
        return open(made).read()
    out = os.path.join(str(tmp_path), src + ".s")
    subprocess.run([HIPCC, *_makefile_flags(), "-I", CSRC, "-S", "--cuda-device-only", os.path.join(CSRC, src), "-o", out], check=True,
                   stderr=subprocess.DEVNULL)
    return open(out).read()


def _kernels(asm):
    """name -> (instruction lines after the compatibility header, preload length)"""
    res = {}
    for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)\.amdhsa_user_sgpr_kernarg_preload_length (\d+)", asm, re.S | re.M):
        body = [l.strip() for l in m.group(2).split("\n")]
        body = [l for l in body if l and not l.startswith((";", ".", "_Z"))]
        res[m.group(1)] = (body, int(m.group(3)))
    return res


pytestmark = pytest.mark.skipif(not os.path.exists(HIPCC) or shutil.which("c++filt") is None, reason="needs hipcc")


def test_makefile_asks_for_kernarg_preload():
    assert "-amdgpu-kernarg-preload-count=16" in " ".join(_makefile_flags())


def test_decode_attention_merge_takes_its_arguments_preloaded(tmp_path):
    ks = _kernels(_asm("k_decattn.hip", tmp_path))
    merge = {n: v for n, v in ks.items() if "dec_attn_merge_kernel" in n}
    assert merge
    for n, (body, pre) in merge.items():
        assert pre >= 13, (n, pre)
        # behind the compatibility header (s_load ... s_branch) the kernel reads the argument segment at most for the optional
        # device-side step counter: no s_load from s[0:1] (the segment pointer) before the first global load
        start = next(i for i, l in enumerate(body) if l.startswith("s_branch")) + 1
        first_ld = next(i for i, l in enumerate(body) if l.startswith("global_load"))
        assert not [l for l in body[start:first_ld] if l.startswith("s_load") and "s[0:1]" in l], n


def test_fast_decode_projections_request_weights_before_any_argument_fetch(tmp_path):
    ks = _kernels(_asm("k_decgemm.hip", tmp_path))
    # the FAST instantiations of the decode step: <MODE, NORM, ACT, OUTF32, NW, NL, WIDE = 0, PK = 1, FAST = 1>
    fast = {n: v for n, v in ks.items() if re.search(r"dec_gemm2_kernelILi\dELi\dELi\dELi\dELi\d+ELi\d+ELi0ELi1ELi1E", n)}
    assert len(fast) >= 8
    for n, (body, pre) in fast.items():
        assert pre >= 14, (n, pre)
        start = next(i for i, l in enumerate(body) if l.startswith("s_branch")) + 1
        first_ld = next(i for i, l in enumerate(body) if l.startswith("global_load"))
        seg_loads = [i for i, l in enumerate(body[start:], start) if l.startswith("s_load") and "s[0:1]" in l]
        assert first_ld - start <= 40, (n, first_ld - start)                 # first request within 40 instructions of entry
        if re.search(r"ELi0ELi1ELi1ELi1EEE", n):
            # fp8 form (F8 = 1): the row-scale pointer lives in the record, so ONE scalar fetch may be issued early — but nothing may WAIT
            # for the segment before the tile's weight requests (the `nt` loads) are out
            waits = [i for i, l in enumerate(body[start:], start) if l.startswith("s_waitcnt") and "lgkmcnt" in l]
            nt = [i for i, l in enumerate(body[start:], start) if l.startswith("global_load") and l.rstrip().endswith(" nt")]
            assert nt and (not waits or waits[0] > nt[0]), n
            assert len([i for i in seg_loads if i < first_ld]) <= 1, n
            continue
        assert not seg_loads or seg_loads[0] > first_ld, n                   # ... and before anything is fetched from the segment


def test_four_wave_gemm_keeps_its_tile_in_registers_and_its_memory_instructions_inside_the_mfma_clusters(tmp_path):
    """gemm_w4_kernel (k_gemm.hip): one wave per SIMD owns a 16 WM x 16 WN tile — WM x WN x 4 accumulator registers (AGPRs), no scratch —
    and, in the steady-state K loop, the LDS-DMA pieces and fragment reads of a K-tile are issued BETWEEN the MFMAs of its clusters
    (sched_group_barrier); placed between the clusters instead, the same kernel measured 20 - 30 % slower (NOTES.md)"""
    asm = _asm("k_gemm.hip", tmp_path)
    found = 0
    for m in re.finditer(r"^(_ZN3bra14gemm_w4_kernelILi(\d)ELi(\d)ELi(\d)EEEvNS_8GemmArgsE):(.*?)\.Lfunc_end.*?; NumAgprs: (\d+).*?; ScratchSize: (\d+)",
                         asm, re.S | re.M):
        wm, wn, body, agprs, scratch = int(m.group(3)), int(m.group(4)), m.group(5), int(m.group(6)), int(m.group(7))
        found += 1
        assert scratch == 0, (m.group(1), scratch)
        assert agprs >= 4 * wm * wn, (m.group(1), agprs)
        ops = []
        for l in body.split("\n"):
            l = l.strip()
            if not l or l.startswith((";", ".")):
                continue
            op = l.split()[0]
            ops.append("M" if op.startswith("v_mfma") else "r" if op.startswith("ds_read") else "D" if "load_lds" in op else
                       "|" if op.startswith("s_barrier") else ".")
        t = "".join(ops)
        # some stretch between two barriers holds a whole cluster pair's worth of MFMAs with every DMA piece and >= WM + WN reads among them
        ok = False
        for seg in t.split("|"):
            first, last = seg.find("M"), seg.rfind("M")
            if first < 0:
                continue
            inner = seg[first:last + 1]
            if inner.count("M") >= wm * wn and inner.count("D") >= wm + wn and seg[first:].count("r") >= wm + wn and inner.count("r") >= wm + wn - 1:
                ok = True                     # (the last read may follow the cluster's last MFMA)
        assert ok, m.group(1)
    assert found == 8, found                   # (EPI_BF16, EPI_F32) x four tile configurations
