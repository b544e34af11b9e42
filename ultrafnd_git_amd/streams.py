"""HIP streams confined to a share of the chip's compute units (csrc/runtime.hip).

The step runs three concurrent chains -- text encoder, visual encoder, head -> exchange -> optimizer -- whose
GEMM workgroups each own a whole CU (112-144 KiB of LDS).  On ordinary streams the chains displace each other
CU by CU: the text chain alone takes 1.45 ms, beside the visual chain 2.1 ms.  With a CU mask per stream each
chain keeps its share: (text, visual) = (192, 64) CUs by default, 24 + 8 CUs of every XCD, so that every XCD's
L2 serves both chains and a text GEMM of 192 tiles is exactly one round on its share.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Sequence

import torch

from . import _lib as L

# How bit b of a CU mask maps to hardware on gfx950 (measured with tools/cu_mask_probe.py):
#   "striped": bit b -> XCD b % 8, CU b // 8 of that XCD;  "block": bit b -> XCD b // 32, CU b % 32.
LAYOUT = "striped"
XCDS, CUS_PER_XCD = 8, 32


def partition_bits(shares: Sequence[int], layout: str = None) -> List[List[int]]:
    """Split the 256 CUs into len(shares) disjoint sets of shares[i] CUs, each an equal slice of every XCD.
    Returns the mask bit numbers of each set."""
    layout = layout or LAYOUT
    if sum(shares) > XCDS * CUS_PER_XCD or any(s <= 0 or s % XCDS for s in shares):
        raise ValueError(f"CU shares {tuple(shares)} must be positive multiples of {XCDS} summing to <= {XCDS * CUS_PER_XCD}")
    out, lo = [], 0
    for s in shares:
        per = s // XCDS
        cus = range(lo, lo + per)                      # CU indices inside every XCD
        if layout == "striped":
            out.append(sorted(c * XCDS + x for c in cus for x in range(XCDS)))
        elif layout == "block":
            out.append(sorted(x * CUS_PER_XCD + c for c in cus for x in range(XCDS)))
        else:
            raise ValueError(f"unknown CU mask layout {layout!r}")
        lo += per
    return out


def mask_words(bits: Sequence[int], n_words: int = 8):
    words = (C.c_uint32 * n_words)()
    for b in bits:
        if not 0 <= b < 32 * n_words:
            raise ValueError(f"CU bit {b} outside the mask")
        words[b // 32] |= 1 << (b % 32)
    return words


class MaskedStream(torch.cuda.ExternalStream):
    """torch view of a hipStream_t created by ufnd_stream_create_cu_mask (close() destroys it)."""

    def __new__(cls, device: torch.device, bits: Sequence[int]):
        words = mask_words(bits)
        out = C.c_void_p()
        with torch.cuda.device(device):
            L.check(L.lib().ufnd_stream_create_cu_mask(words, len(words), C.byref(out)), "ufnd_stream_create_cu_mask")
        self = super().__new__(cls, out.value, device=device)
        self._raw, self.cu_bits = out.value, tuple(bits)
        return self

    def close(self) -> None:
        """Destroy the stream (it must be idle and no captured graph may still name it).  Streams that are never
        closed live as long as the process: tearing a stream down from a finalizer at interpreter exit, after the
        allocator and graphs that reference it, faults inside the HIP runtime."""
        raw, self._raw = getattr(self, "_raw", None), None
        if raw:
            L.check(L.lib().ufnd_stream_destroy(raw), "ufnd_stream_destroy")
