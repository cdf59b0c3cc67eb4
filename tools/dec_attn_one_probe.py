"""decode attention alone at the cfg-3 shape (1 prompt x 8 rollouts, P = 2180, Hq/Hkv = 16/8, hd 128), 28 layers' worth of
distinct caches per pass: us per layer of the first generation (bra_dec_attn_both + bra_attn_decode_merge) vs k_decattn.hip
(bra_dec_attn_one: items kernel + merge kernel).  PROBE_T = completion tokens already cached."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bioreason_amd._lib import get_lib, current_stream

dev = torch.device("cuda:0")
BF = torch.bfloat16
L, R, copies, Hq, Hkv, hd, P, C = 28, 1, 8, 16, 8, 128, 2180, 256
t = int(os.environ.get("PROBE_T", "200"))
B, Nq, Nkv = R * copies, Hq * hd, Hkv * hd
pitch, cp = (P + 63) // 64 * 64, (C + 63) // 64 * 64
g = torch.Generator(device=dev).manual_seed(1)
rn = lambda *s: torch.randn(*s, generator=g, device=dev).to(BF)
qkv = rn(B, Nq + 2 * Nkv)
qw, kw = torch.ones(hd, dtype=BF, device=dev), torch.ones(hd, dtype=BF, device=dev)
kp = [rn(R, Hkv, P, hd) for _ in range(L)]
vtp = [rn(R, Hkv, hd, pitch) for _ in range(L)]
kc = [rn(B, Hkv, C, hd) for _ in range(L)]
vc = [rn(B, Hkv, C, hd) for _ in range(L)]
vct = [v.transpose(2, 3).contiguous() for v in vc]
npos = P + C + 1
inv = 1.0 / (10000.0 ** (torch.arange(0, hd, 2).float() / hd))
ang = torch.arange(npos).float()[:, None] * inv[None]
cosT, sinT = ang.cos().to(dev).contiguous(), ang.sin().to(dev).contiguous()
pos = torch.full((B,), P + t, dtype=torch.int32, device=dev)
rope_rows = torch.cat([cosT[pos.long()], sinT[pos.long()]], -1).contiguous()
nslot = (P + 63) // 64 + (C + 63) // 64 + 1
part_o = torch.zeros(B * Hq, nslot, hd, dtype=torch.float32, device=dev)
part_ml = torch.zeros(B * Hq, nslot, 2, dtype=torch.float32, device=dev)
o = torch.zeros(B, Nq, dtype=BF, device=dev)
lib = get_lib()
st = current_stream(qkv)
eps, scale = 1e-6, hd ** -0.5
npc = (P + 63) // 64


def run_both():
    for li in range(L):
        lib.call("bra_dec_attn_both", qkv, Nq + 2 * Nkv, qw, kw, cosT, sinT, pos, kp[li], Hkv * P * hd, P * hd, hd, vtp[li], Hkv * hd * pitch,
                 hd * pitch, pitch, None, kc[li], vc[li], part_o, part_ml, R, copies, Hq, Hkv, hd, P, C, t, eps, scale, None, rope_rows, st)
        lib.call("bra_attn_decode_merge", part_o, part_ml, o, B, Hq, hd, npc + (t + 64) // 64, None, npc, st)


def run_one():
    for li in range(L):
        lib.call("bra_dec_attn_one", qkv, Nq + 2 * Nkv, qw, kw, cosT, sinT, pos, rope_rows, kp[li], Hkv * P * hd, P * hd, hd, vtp[li],
                 Hkv * hd * pitch, hd * pitch, pitch, None, kc[li], vct[li], cp, part_o, part_ml, nslot, o, Nq, R, copies,
                 Hq, Hkv, hd, P, C, t, eps, scale, None, st)


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps / L * 1e3


run_both(); o_both = o.clone()
run_one(); torch.cuda.synchronize()
print("max |one - both| = %.4f  (|both| max %.3f)" % ((o.float() - o_both.float()).abs().max().item(), o_both.float().abs().max().item()))
print("t=%d  both+merge %.2f us/layer   items+merge (k_decattn) %.2f us/layer" % (t, timeit(run_both), timeit(run_one)), flush=True)
