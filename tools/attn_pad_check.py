import os, sys, torch
sys.path.insert(0, '/root/repo' if os.path.isdir('/root/repo/bioreason_amd') else os.getcwd())
from bioreason_amd import ops, _lib
lib = _lib.use_debug_library()
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
def rnd(*s): return (torch.randn(*s, generator=g) * 0.5).to(torch.bfloat16).to(dev)
for (B, S, Hq, Hkv, hd, pads) in ((2, 2180, 16, 8, 128, [0, 300]), (2, 464, 32, 8, 128, [0, 37]), (3, 1000, 4, 2, 128, [5, 0, 130])):
    q, k, v = rnd(B, S, Hq, hd), rnd(B, S, Hkv, hd), rnd(B, S, Hkv, hd)
    kmask = torch.ones(B, S, dtype=torch.uint8, device=dev)
    for b, p in enumerate(pads): kmask[b, :p] = 0
    vt = ops.head_transpose(v)
    res = {}
    for on in (1, 0):
        lib.call("bra_attn_set_fwd4", on)
        res[on] = ops.attn_fwd(q, k, vt, kmask, True, hd ** -0.5, nsplit=1)
    o4, o8 = res[1][0].float(), res[0][0].float()
    qf, kf, vf = q.float(), k.float().repeat_interleave(Hq // Hkv, 2), v.float().repeat_interleave(Hq // Hkv, 2)
    sc = torch.einsum("bqhd,bkhd->bhqk", qf, kf) * hd ** -0.5
    vis = (torch.arange(S, device=dev)[None, :] <= torch.arange(S, device=dev)[:, None])[None, None] & kmask.bool()[:, None, None, :]
    sc = sc.masked_fill(~vis, float("-inf"))
    ref = torch.einsum("bhqk,bkhd->bqhd", torch.nan_to_num(sc.softmax(-1), nan=0.0), vf)
    valid = kmask.bool()[:, :, None, None]
    e4 = ((o4 - ref).abs() * valid).amax(dim=(2, 3)); e8 = ((o8 - ref).abs() * valid).amax(dim=(2, 3))
    print(f"B{B} S{S} Hq{Hq}: max abs err per (b,q): new max {e4.max().item():.4f} mean {e4.mean().item():.5f} | old max {e8.max().item():.4f} mean {e8.mean().item():.5f}; worst q new {int(e4.argmax()) % S} old {int(e8.argmax()) % S}")
    rn = (((o4 - ref) * valid).norm(dim=(2, 3)) / ((ref * valid).norm(dim=(2, 3)) + 1e-9)); ro = (((o8 - ref) * valid).norm(dim=(2, 3)) / ((ref * valid).norm(dim=(2, 3)) + 1e-9))
    print(f"     per-row rel err: new max {rn.max().item():.4f} p99 {rn.flatten().quantile(0.99).item():.4f} mean {rn.mean().item():.4f} | old max {ro.max().item():.4f} p99 {ro.flatten().quantile(0.99).item():.4f} mean {ro.mean().item():.4f}")
    lse4, lse8 = res[1][1], res[0][1]
    lref = torch.logsumexp(sc, -1)
    ok = torch.isfinite(lref)
    print(f"     lse max abs err new {((lse4 - lref).abs()[ok]).max().item():.2e} old {((lse8 - lref).abs()[ok]).max().item():.2e}")
