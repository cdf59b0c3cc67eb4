"""Static view of k_attn4.hip's code for gfx950 (no GPU needed): registers / spills per kernel and, for the hot loop (the innermost
loop that holds 64 / 32 MFMAs), the instruction mix of every {MFMA .. next MFMA} gap.
  python tools/attn4_isa.py [extra hipcc flags ...]"""
import collections, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "bioreason_amd", "csrc")
OUT = "/tmp/attn4_isa.s"
flags = "--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffast-math -fno-finite-math-only -Wno-unused-result -I. -mllvm -amdgpu-kernarg-preload-count=16".split()
src = os.environ.get("A4_SRC", "k_attn4.hip")
subprocess.run(["/opt/rocm/bin/hipcc", *flags, *sys.argv[1:], "-S", "--cuda-device-only", src, "-o", OUT], cwd=CSRC, check=True,
               stderr=subprocess.DEVNULL)
txt = open(OUT).read()
kernels = re.split(r"\n(?=_ZN3bra\w+: )", txt)
def cls(ins):
    op = ins.split()[0]
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith("v_accvgpr"): return "acc"
    if op.startswith("v_exp"): return "exp"
    if op.startswith("v_"): return "valu"
    if op.startswith("ds_"): return "ds"
    if op.startswith("global_load_lds") or op.startswith("buffer_load") and "lds" in ins: return "dma"
    if op.startswith("scratch_"): return "scratch"
    if op.startswith("global_") or op.startswith("buffer_"): return "vmem"
    if op.startswith("s_waitcnt"): return "wait"
    if op.startswith("s_nop"): return "nop"
    if op.startswith("s_barrier"): return "bar"
    if op.startswith("s_cbranch") or op.startswith("s_branch"): return "br"
    if op.startswith("s_"): return "salu"
    return "other"
for k in kernels[1:]:
    name = k.split(":", 1)[0]
    meta = {m: re.search(rf"; {m}: (\d+)", k) for m in ("NumVgprs", "NumAgprs", "ScratchSize", "Occupancy")}
    lines = [l.strip() for l in k.split("\n")]
    ins = [l for l in lines if l and not l.startswith((";", ".", "_Z")) and not l.endswith(":") and not l.startswith("s_endpgm")]
    tot = collections.Counter(cls(i) for i in ins)
    print(f"== {name}\n   " + " ".join(f"{m}={v.group(1)}" for m, v in meta.items() if v) + f"  total: {dict(tot)}")
    # innermost loops: label ... s_cbranch back to label
    labels = {}
    seq = []
    for l in lines:
        if re.match(r"^\.LBB\d+_\d+:", l): labels[l.split(":")[0]] = len(seq)
        elif l and not l.startswith((";", ".")) and not l.endswith(":"): seq.append(l)
    loops = []
    for i, l in enumerate(seq):
        m = re.match(r"s_cbranch_\w+ (\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] <= i: loops.append((labels[m.group(1)], i))
    for lo, hi in loops:
        body = seq[lo:hi + 1]
        c = collections.Counter(cls(i) for i in body)
        if c["mfma"] < 16: continue
        print(f"   loop [{lo}:{hi}] {len(body)} instructions: {dict(c)}")
        gaps, cur = [], None
        for i in body:
            if cls(i) == "mfma":
                if cur is not None: gaps.append(cur)
                cur = collections.Counter()
            elif cur is not None: cur[cls(i)] += 1
        if os.environ.get("A4_GAPS"):
            for n, g in enumerate(gaps): print(f"      gap {n:3d}: {sum(g.values()):3d}  {dict(g)}")
        tots = [sum(g.values()) for g in gaps]
        print(f"      gaps: n={len(gaps)} mean={sum(tots)/max(1,len(tots)):.1f} max={max(tots)} min={min(tots)}")
