"""Per-device cogs handle and grow-only workspaces (torch owns the device memory)."""
from __future__ import annotations

import ctypes as C
from typing import Dict

import torch

from . import _lib as L


class Handle:
    def __init__(self, device_index: int):
        self.device_index = device_index
        self._h = C.c_void_p()
        L.check(L.lib.cogs_create(device_index, C.byref(self._h)), "cogs_create")
        self._ws: Dict[str, torch.Tensor] = {}
        self._active: Dict[str, object] = {}   # which packed weight set ("vit" / "proj" / "llm") the handle points at

    @property
    def h(self):
        return self._h

    def workspace(self, name: str, nbytes: int) -> torch.Tensor:
        cur = self._ws.get(name)
        if cur is None or cur.numel() < nbytes:
            self._ws[name] = None  # drop the old buffer before growing
            cur = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=f"cuda:{self.device_index}")
            self._ws[name] = cur
        return cur

    def activate(self, kind: str, owner, load) -> None:
        """the handle keeps ONE pointer table per model kind; an engine re-points it before use when another
        engine (a second adapter, another test model) was loaded in between"""
        if self._active.get(kind) is not owner:
            load()
            self._active[kind] = owner

    def close(self):
        if self._h:
            L.lib.cogs_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_HANDLES: Dict[int, Handle] = {}


def get_handle(device) -> Handle:
    dev = torch.device(device)
    if dev.type != "cuda":
        raise L.CogsError("cogstream_amd runs on ROCm devices only (got %s)" % dev)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    if idx not in _HANDLES:
        _HANDLES[idx] = Handle(idx)
    return _HANDLES[idx]
