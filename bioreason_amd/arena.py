"""Flat storage for everything that trains (LoRA adapters + dna_projection).

One fp32 master buffer, one fp32 gradient buffer, AdamW moments, and a byte mask of structural zeros; the
nn.Parameters the reference's callers see (`...lora_A.default.weight`, `dna_projection.weight`, ...) are VIEWS
into the master buffer and their `.grad` are views into the gradient buffer.  Consequences on MI355X:
  * the data-parallel gradient reduction is ONE RCCL all-reduce over one contiguous bucket (≈148 MB fp32 for
    Qwen3-1.7B r=32), instead of the reference's DDP / ZeRO-2 bucket machinery;
  * AdamW + global-norm clip is one fused launch over the bucket;
  * the bf16 working images the GEMMs read (fused/transposed LoRA factors) are refreshed by one
    table-driven pack launch per optimiser step.
The arena is append-only: `add()` new blocks, then `commit()` (re)allocates, keeps the old values and calls
the owners' rebind callbacks so their Parameters alias the new storage.
"""
from __future__ import annotations

import ctypes
from typing import Callable, Dict, List, Optional, Tuple

import torch

from . import ops

BF16 = torch.bfloat16


class _PackDesc(ctypes.Structure):
    _fields_ = [("src", ctypes.c_void_p), ("src_ld", ctypes.c_long), ("dst", ctypes.c_void_p), ("dst_ld", ctypes.c_long),
                ("rows", ctypes.c_int), ("cols", ctypes.c_int), ("transpose", ctypes.c_int), ("pad", ctypes.c_int)]


class TrainableArena:
    def __init__(self, device):
        self.device = torch.device(device)
        self._shapes: Dict[str, Tuple[int, int]] = {}
        self._offsets: Dict[str, int] = {}
        self._next = 0
        self.numel = 0
        self.params: Optional[torch.Tensor] = None
        self.grads: Optional[torch.Tensor] = None
        self.mask: Optional[torch.Tensor] = None
        self.exp_avg: Optional[torch.Tensor] = None
        self.exp_avg_sq: Optional[torch.Tensor] = None
        self._rebind: List[Callable[[], None]] = []
        self._packs: List[Tuple[torch.Tensor, torch.Tensor, bool]] = []
        self._pack_table: Optional[torch.Tensor] = None
        self._pack_max = 0
        self.step_count = 0
        self._sumsq: Optional[torch.Tensor] = None
        self._sumsq_ws: Optional[torch.Tensor] = None
        # autograd anchor: a leaf that requires grad, passed into the fused Functions so that their backward
        # (which writes LoRA / projection gradients into `grads` as a side effect) always runs
        self.anchor = torch.zeros(1, dtype=torch.float32, device=self.device, requires_grad=True)

    # ---- layout -------------------------------------------------------------------------------
    def add(self, name: str, rows: int, cols: int) -> None:
        assert name not in self._shapes
        self._shapes[name] = (rows, cols)
        self._offsets[name] = self._next
        self._next += (rows * cols + 63) // 64 * 64

    def on_rebind(self, fn: Callable[[], None]) -> None:
        self._rebind.append(fn)

    def commit(self) -> None:
        if self._next == self.numel and self.params is not None:
            return
        old = (self.params, self.grads, self.mask, self.exp_avg, self.exp_avg_sq, self.numel)
        n = self._next
        self.params = torch.zeros(n, dtype=torch.float32, device=self.device)
        self.grads = torch.zeros(n, dtype=torch.float32, device=self.device)
        self.mask = torch.zeros(n, dtype=torch.uint8, device=self.device)
        if old[0] is not None:
            k = old[5]
            self.params[:k].copy_(old[0]); self.grads[:k].copy_(old[1]); self.mask[:k].copy_(old[2])
            if old[3] is not None:
                self.exp_avg = torch.zeros_like(self.params); self.exp_avg[:k].copy_(old[3])
                self.exp_avg_sq = torch.zeros_like(self.params); self.exp_avg_sq[:k].copy_(old[4])
        self.numel = n
        self._packs, self._pack_table = [], None
        for fn in self._rebind:
            fn()

    def to(self, device) -> None:
        device = torch.device(device)
        if device == self.device:
            return
        self.device = device
        for nm in ("params", "grads", "mask", "exp_avg", "exp_avg_sq"):
            t = getattr(self, nm)
            if t is not None:
                setattr(self, nm, t.to(device))
        self.anchor = torch.zeros(1, dtype=torch.float32, device=device, requires_grad=True)
        self._sumsq = None
        self._sumsq_ws = None
        self._packs, self._pack_table = [], None
        for fn in self._rebind:
            fn()

    def _view(self, buf: torch.Tensor, name: str) -> torch.Tensor:
        r, c = self._shapes[name]
        o = self._offsets[name]
        return buf[o:o + r * c].view(r, c)

    def param(self, name: str) -> torch.Tensor:
        return self._view(self.params, name)

    def grad(self, name: str) -> torch.Tensor:
        return self._view(self.grads, name)

    def mask_view(self, name: str) -> torch.Tensor:
        return self._view(self.mask, name)

    # ---- bf16 working images --------------------------------------------------------------------
    def register_pack(self, src_fp32: torch.Tensor, dst_bf16: torch.Tensor, transpose: bool = False) -> None:
        """dst (bf16) is refreshed from src (fp32 view of the arena) by `pack()`."""
        assert src_fp32.dim() == 2 and dst_bf16.dim() == 2 and src_fp32.stride(1) == 1 and dst_bf16.stride(1) == 1
        self._packs.append((src_fp32, dst_bf16, transpose))
        self._pack_table = None

    def pack_if_stale(self) -> None:
        """refresh the bf16 images after torch-side writes to the master (load_state_dict, manual init)"""
        if self.params is not None and getattr(self, "_packed_version", None) != self.params._version:
            self.pack()

    def pack(self) -> None:
        if self.params is not None:
            self._packed_version = self.params._version
        if not self._packs:
            return
        if self._pack_table is None:
            n = len(self._packs)
            arr = (_PackDesc * n)()
            mx = 0
            for i, (s, d, t) in enumerate(self._packs):
                arr[i] = _PackDesc(s.data_ptr(), s.stride(0), d.data_ptr(), d.stride(0), s.shape[0], s.shape[1], int(t), 0)
                mx = max(mx, s.numel())
            raw = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).clone()
            self._pack_table = raw.to(self.device)
            self._pack_max = mx
        ops.pack_params(self._pack_table, len(self._packs), self._pack_max)

    # ---- optimiser -------------------------------------------------------------------------------
    def zero_grad(self) -> None:
        self.grads.zero_()

    def adamw_step(self, lr: float, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.01,
                   max_grad_norm: float = 0.0, grad_scale: float = 1.0) -> None:
        if self.exp_avg is None:
            self.exp_avg = torch.zeros_like(self.params)
            self.exp_avg_sq = torch.zeros_like(self.params)
        if self._sumsq is None:
            self._sumsq = torch.zeros(1, dtype=torch.float32, device=self.device)
        self.step_count += 1
        ss = None
        if max_grad_norm and max_grad_norm > 0:
            if self._sumsq_ws is None:
                self._sumsq_ws = torch.empty((1024,), dtype=torch.float32, device=self.device)
            ops.sumsq(self.grads, self._sumsq, mask=self.mask, ws=self._sumsq_ws)
            ss = self._sumsq
        ops.adamw(self.params, self.grads, self.exp_avg, self.exp_avg_sq, lr, betas[0], betas[1], eps, weight_decay,
                  self.step_count, sumsq_t=ss, max_norm=max_grad_norm, grad_scale=grad_scale, mask=self.mask)
        self.pack()

    def grad_norm(self) -> torch.Tensor:
        ss = torch.zeros(1, dtype=torch.float32, device=self.device)
        ops.sumsq(self.grads, ss, mask=self.mask)
        return ss.sqrt()
