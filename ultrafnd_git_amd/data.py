"""The data side of the trainer: the tensorised split of the feature cache, resident in HBM, and the loader over it
(the reference's `CachedTensorDataset` + `DataLoader(..., num_workers=0)` + default collate, src/training/forensic_trainer.py:60-83,
227-234), plus the FakeSV-shaped synthetic cache the tests and the benchmark use.

A batch is an index gather on the device: the loader yields `IndexedBatch`es -- the dict default collate would build, keyed
the same, gathered lazily -- and the step fills its static buffers from the indices in one `ufnd_gather_rows` launch."""
from __future__ import annotations

from typing import Dict, Iterator, Optional

import numpy as np
import torch

from .dp import shard_indices, world_info

class CachedTensorDataset(torch.utils.data.Dataset):
    """Tensorised view of one split of the cache (forensic_trainer.py:60-83), device-resident."""

    def __init__(self, cache: Dict, indices: np.ndarray, device: Optional[torch.device] = None):
        indices = np.asarray(indices, dtype=np.int64)
        self.ids = cache["ids"][indices] if "ids" in cache else indices
        self.global_idx = torch.from_numpy(indices)

        def take(key, dtype):
            a = cache[key]
            t = a if isinstance(a, torch.Tensor) else torch.from_numpy(np.asarray(a))
            t = t[torch.from_numpy(indices).to(t.device)].to(dtype)
            return t.to(device) if device is not None else t
        self.T = take("text", torch.float32)
        self.A = take("audio", torch.float32)
        self.V = take("visual", torch.float32)
        self.U = take("temporal", torch.float32)
        self.AUX = take("aux", torch.float32)
        self.y = take("labels", torch.int64)
        self.G = take("gnn_Z", torch.float32) if "gnn_Z" in cache else None
        # raw inputs for encode_inline
        self.ids_tok = take("input_ids", torch.int64) if "input_ids" in cache else None
        self.mask_tok = take("attention_mask", torch.int32) if "attention_mask" in cache else None
        self.frames = take("frames", torch.float32) if "frames" in cache else None
        if device is not None:
            self.global_idx = self.global_idx.to(device)

    def __len__(self):
        return self.T.shape[0]

    def __getitem__(self, i):
        return {"text_features": self.T[i], "audio_features": self.A[i], "visual_features": self.V[i],
                "temporal_features": self.U[i], "aux": self.AUX[i], "label": self.y[i], "index": i}

    def gather(self, idx: torch.Tensor) -> Dict[str, torch.Tensor]:
        """Default-collate equivalent for a whole index vector, as one gather per tensor."""
        b = {"text_features": self.T[idx], "audio_features": self.A[idx], "visual_features": self.V[idx],
             "temporal_features": self.U[idx], "aux": self.AUX[idx], "label": self.y[idx], "index": idx}
        if self.ids_tok is not None:
            b["input_ids"], b["attention_mask"] = self.ids_tok[idx], self.mask_tok[idx]
        if self.frames is not None:
            b["frames"] = self.frames[idx]
        return b


class IndexedBatch(dict):
    """A batch of a device-resident split, named by its row indices (`batch["index"]`).  It is the dict the reference's
    default collate would build (same keys); a tensor is gathered when it is first asked for.  The trainer never asks:
    it sends the indices to `ufnd_gather_rows`, which fills the step's static buffers in one launch."""
    _SRC = {"text_features": "T", "audio_features": "A", "visual_features": "V", "temporal_features": "U", "aux": "AUX",
            "label": "y", "input_ids": "ids_tok", "attention_mask": "mask_tok", "frames": "frames"}

    def __init__(self, ds: "CachedTensorDataset", idx: torch.Tensor):
        super().__init__(index=idx)
        self.ds = ds

    def _lazy(self, k) -> bool:
        return k in self._SRC and getattr(self.ds, self._SRC[k]) is not None

    def __missing__(self, k):
        if not self._lazy(k):
            raise KeyError(k)
        v = getattr(self.ds, self._SRC[k])[dict.__getitem__(self, "index")]
        self[k] = v
        return v

    def __contains__(self, k):
        return dict.__contains__(self, k) or self._lazy(k)

    def get(self, k, default=None):
        return self[k] if k in self else default

    def keys(self):
        return [k for k in self._SRC if self._lazy(k)] + [k for k in dict.keys(self) if k not in self._SRC]

    def __iter__(self):
        return iter(self.keys())

    def __len__(self):
        return len(self.keys())

    def items(self):
        return [(k, self[k]) for k in self.keys()]

    def values(self):
        return [self[k] for k in self.keys()]


def _batch_size(batch) -> int:
    if type(batch) is IndexedBatch:
        return int(dict.__getitem__(batch, "index").numel())
    return int(batch["label"].shape[0])


class DeviceBatchLoader:
    """DataLoader(dataset, batch_size, shuffle, drop_last=False) over a device-resident split,
    sharded across data-parallel ranks (DistributedSampler semantics)."""

    def __init__(self, dataset: CachedTensorDataset, batch_size: int, shuffle: bool, seed: int = 0, group=None, pad: Optional[bool] = None):
        self.dataset, self.batch_size, self.shuffle, self.seed, self.group = dataset, int(batch_size), shuffle, seed, group
        # training shards are wrapped to equal length (every rank takes the same number of steps: one collective per
        # step); evaluation shards are not, so that no sample enters the epoch metrics twice
        self.pad = shuffle if pad is None else pad
        self.epoch = 0

    def _indices(self) -> torch.Tensor:
        n = len(self.dataset)
        world, rank = world_info(self.group)
        perm = None
        if self.shuffle:
            g = torch.Generator().manual_seed(self.seed + self.epoch)
            perm = torch.randperm(n, generator=g)
        return shard_indices(n, world, rank, perm, pad=self.pad)

    def __len__(self):
        n = self._indices().numel()
        return (n + self.batch_size - 1) // self.batch_size

    def __iter__(self) -> Iterator[Dict[str, torch.Tensor]]:
        idx = self._indices().to(self.dataset.T.device)
        self.epoch += 1
        for s in range(0, idx.numel(), self.batch_size):
            yield IndexedBatch(self.dataset, idx[s:s + self.batch_size])



# ---------------------------------------------------------------------------------------------
def synthetic_cache(n: int, seed: int = 0, gnn_dim: int = 128, with_raw: bool = False, seq_len: int = 128,
                    frames: int = 1, vocab: int = 30522) -> Dict:
    """FakeSV-shaped synthetic cache: the reference's own smoke test feeds randn
    features (scripts/smoke_test_v2.py:43-45).  70/15/15 split."""
    g = torch.Generator().manual_seed(seed)

    def l2(x):
        return x / x.norm(dim=1, keepdim=True)
    cache = {"ids": np.array([f"syn{i}" for i in range(n)]), "labels": torch.randint(0, 2, (n,), generator=g).numpy(),
             "text": l2(torch.randn(n, 768, generator=g)).numpy(), "audio": l2(torch.randn(n, 128, generator=g)).numpy(),
             "visual": l2(torch.randn(n, 512, generator=g)).numpy(), "temporal": torch.randn(n, 256, generator=g).numpy(),
             "aux": torch.rand(n, 2, generator=g).numpy(), "gnn_Z": torch.randn(n, gnn_dim, generator=g).numpy()}
    perm = torch.randperm(n, generator=g).numpy()
    a, b = int(0.7 * n), int(0.85 * n)
    cache["split"] = (np.sort(perm[:a]), np.sort(perm[a:b]), np.sort(perm[b:]))
    if with_raw:
        ids = torch.randint(0, vocab, (n, seq_len), generator=g)
        ids[:, 0] = min(101, vocab - 1)
        lens = torch.randint(min(16, seq_len), seq_len + 1, (n,), generator=g)
        cache["input_ids"] = ids.numpy()
        cache["attention_mask"] = (torch.arange(seq_len)[None] < lens[:, None]).to(torch.int32).numpy()
        cache["frames"] = torch.randn(n, frames, 3, 224, 224, generator=g).numpy()
    return cache
