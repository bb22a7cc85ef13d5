"""Minimal ragged point-cloud batch for the oracle (test infrastructure).

Mirrors only what the hot path needs of reference structures/pointclouds.py: per-batch lists of
(N_b, C) tensors, the zero-padded (B, max N_b, C) view (structures/structutils.py:47-86), the
non-pad mask (:790-809) and append (:1203-1235).  `feats` holds the confidence counts (C=1)."""
from dataclasses import dataclass, field
from typing import List, Optional

import torch


def _pad(xs: List[torch.Tensor], n: int) -> torch.Tensor:
    out = torch.zeros((len(xs), n, xs[0].shape[1]), dtype=xs[0].dtype, device=xs[0].device)
    for b, x in enumerate(xs):
        if len(x):
            out[b, : x.shape[0]] = x
    return out


@dataclass
class Cloud:
    points: Optional[List[torch.Tensor]] = None
    normals: Optional[List[torch.Tensor]] = None
    colors: Optional[List[torch.Tensor]] = None
    feats: Optional[List[torch.Tensor]] = None

    @property
    def has_points(self) -> bool:
        return self.points is not None

    def __len__(self) -> int:
        return 0 if self.points is None else len(self.points)

    @property
    def counts(self) -> List[int]:
        return [int(p.shape[0]) for p in self.points]

    @property
    def nmax(self) -> int:
        return max(self.counts)

    def padded(self, name: str) -> Optional[torch.Tensor]:
        xs = getattr(self, name)
        return None if xs is None else _pad(xs, self.nmax)

    def nonpad_mask(self) -> torch.Tensor:
        m = torch.zeros((len(self), self.nmax), dtype=torch.bool)
        for b, n in enumerate(self.counts):
            m[b, :n] = True
        return m

    def set_from_padded(self, name: str, value: torch.Tensor):
        setattr(self, name, [value[b, :n] for b, n in enumerate(self.counts)])

    def append(self, other: "Cloud") -> "Cloud":
        if not other.has_points:
            return self
        if not self.has_points:
            return Cloud(*[None if x is None else [t.clone() for t in x]
                           for x in (other.points, other.normals, other.colors, other.feats)])
        cat = lambda a, b: None if a is None else [torch.cat([x, y], 0) for x, y in zip(a, b)]
        return Cloud(cat(self.points, other.points), cat(self.normals, other.normals),
                     cat(self.colors, other.colors), cat(self.feats, other.feats))
