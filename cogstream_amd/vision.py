"""Host mirror of the reference's vision seams, computing through the C ABI.

  VisionEncoder(pixel_values=, grid_sizes=, merge_sizes=)   model/cogreasoner_chat.py:270-274
                                                            (Videollama3VisionEncoderModel.forward,
                                                             model/modeling_videollama3_encoder.py:479-510)
  Projector(x)                                              model/cogreasoner_chat.py:275 (mm_projector)
"""
from __future__ import annotations

import ctypes as C
from typing import Dict

import torch

from . import _lib as L
from .runtime import get_handle
from .weights import PackedProjector, PackedVit, VisionConfig

BLOCK_DIAG = L.ATTN_BLOCK_DIAG
REF_EAGER_GLOBAL = L.ATTN_REF_EAGER_GLOBAL


class VisionEncoder:
    def __init__(self, state: Dict[str, torch.Tensor], cfg: VisionConfig, dtype=torch.bfloat16, device="cuda",
                 attn_mode: int = BLOCK_DIAG, fold_ln=None):
        """fold_ln: None = the default of weights.PackedVit (LayerNorm folded into the QKV / fc1 GEMMs in bf16);
        False keeps LayerNorm as its own kernel (A/B runs: bench.py --no-ln-fold)"""
        self.cfg, self.dtype, self.device = cfg, dtype, torch.device(device)
        self.attn_mode = attn_mode
        self.handle = get_handle(self.device)
        self.packed = PackedVit(state, cfg, dtype, self.device, fold_ln=fold_ln)
        self._activate()

    def _activate(self):
        self.handle.activate("vit", self, lambda: L.check(
            L.lib.cogs_vit_load(self.handle.h, C.byref(self.packed.struct)), "cogs_vit_load"))

    def __call__(self, pixel_values: torch.Tensor, grid_sizes: torch.Tensor, merge_sizes: torch.Tensor,
                 attn_mode=None, projector=None) -> torch.Tensor:
        if not pixel_values.is_cuda:
            raise L.CogsError("pixel_values must live on the GPU")
        self._activate()
        pixel_values = pixel_values.contiguous()
        gs = [int(v) for v in grid_sizes.reshape(-1).tolist()]
        ms = [int(v) for v in merge_sizes.reshape(-1).tolist()]
        V = len(ms)
        n = sum(gs[3 * v] * gs[3 * v + 1] * gs[3 * v + 2] for v in range(V))
        m = sum(gs[3 * v] * gs[3 * v + 1] * gs[3 * v + 2] // (ms[v] * ms[v]) for v in range(V))
        if pixel_values.shape != (n, self.cfg.patch_dim):
            raise L.CogsError(f"pixel_values {tuple(pixel_values.shape)} does not match grid_sizes ({n} patches)")
        nbytes = C.c_size_t()
        L.check(L.lib.cogs_vit_workspace_bytes(self.handle.h, n, C.byref(nbytes)), "cogs_vit_workspace_bytes")
        ws = self.handle.workspace("vit", nbytes.value)
        out = torch.empty(m, self.cfg.hidden_size, device=self.device, dtype=self.dtype)
        gsa = (C.c_int64 * len(gs))(*gs)
        msa = (C.c_int64 * len(ms))(*ms)
        mode = self.attn_mode if attn_mode is None else attn_mode
        if projector is None:
            L.check(L.lib.cogs_vit_encode(self.handle.h, L.current_stream(), pixel_values.data_ptr(),
                                          L.dtype_code(pixel_values.dtype), gsa, msa, V, mode, out.data_ptr(),
                                          ws.data_ptr(), ws.numel()), "cogs_vit_encode")
            return out
        # encoder + projector in one call (cogs_vit_encode_project): every frame range projects its own tokens on its stream
        if projector.handle is not self.handle or projector.dtype != self.dtype:
            raise L.CogsError("encode_project: encoder and projector must share the device handle and the dtype")
        projector._activate()
        es = 2 if self.dtype == torch.bfloat16 else 4
        pws = self.handle.workspace("proj", m * projector.packed.out_dim * es)
        pout = torch.empty(m, projector.packed.out_dim, device=self.device, dtype=self.dtype)
        L.check(L.lib.cogs_vit_encode_project(self.handle.h, L.current_stream(), pixel_values.data_ptr(),
                                              L.dtype_code(pixel_values.dtype), gsa, msa, V, mode, out.data_ptr(),
                                              ws.data_ptr(), ws.numel(), pout.data_ptr(), pws.data_ptr(), pws.numel()),
                "cogs_vit_encode_project")
        return out, pout

    def encode_project(self, pixel_values: torch.Tensor, grid_sizes: torch.Tensor, merge_sizes: torch.Tensor, projector,
                       attn_mode=None):
        """(encoder tokens [M, hidden], projected tokens [M, out_dim]) = mm_projector(vision_encoder(...)) of
        cogreasoner_chat.py:270-275 as one library call: the same bits as `projector(self(...))`, with every frame
        range's projection queued on that range's stream"""
        return self(pixel_values, grid_sizes, merge_sizes, attn_mode, projector=projector)


class Projector:
    def __init__(self, state: Dict[str, torch.Tensor], dtype=torch.bfloat16, device="cuda"):
        self.dtype, self.device = dtype, torch.device(device)
        self.handle = get_handle(self.device)
        self.packed = PackedProjector(state, dtype, self.device)
        self._activate()

    def _activate(self):
        self.handle.activate("proj", self, lambda: L.check(
            L.lib.cogs_proj_load(self.handle.h, C.byref(self.packed.struct)), "cogs_proj_load"))

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        self._activate()
        x = x.contiguous()
        M = x.shape[0]
        es = 2 if self.dtype == torch.bfloat16 else 4
        ws = self.handle.workspace("proj", M * self.packed.out_dim * es)
        out = torch.empty(M, self.packed.out_dim, device=self.device, dtype=self.dtype)
        L.check(L.lib.cogs_project(self.handle.h, L.current_stream(), x.data_ptr(), M, out.data_ptr(), ws.data_ptr(),
                                   ws.numel()), "cogs_project")
        return out
