"""Stand-in for chamferdist==1.0.0 `knn_points` (K=1 only): brute force, squared L2,
x->y->z accumulation, first (lowest-index) minimum wins.  Our own code -- see README.md.

With GS_SHIM_KNN=c in the environment the search itself runs in oracle/knn_ref.c (knn1_ref_wide: the same contract,
bit-identical to the torch formulation below -- tests/test_oracle_golden.py compares them) so that the 640x480
golden generators finish in minutes; the default stays the torch brute force."""
import os
import sys
from collections import namedtuple

import torch

_C = None
if os.environ.get("GS_SHIM_KNN") == "c":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
    from oracle import knn as _C  # noqa: E402

_KNN = namedtuple("KNN", "dists idx knn")


def knn_points(p1, p2, lengths1=None, lengths2=None, K=1, version=-1,
               return_nn=False, return_sorted=True):
    assert K == 1 and p1.ndim == 3 and p2.ndim == 3
    dists, idxs = [], []
    for b in range(p1.shape[0]):
        a, t = p1[b].detach(), p2[b].detach()
        if _C is not None:
            _, best_i = _C.knn1(a.contiguous(), t.contiguous(), wide=True)
            g = p2[b][best_i]
            dd = ((p1[b] - g) ** 2)
            dists.append(((dd[:, 0] + dd[:, 1]) + dd[:, 2])[:, None])
            idxs.append(best_i[:, None])
            continue
        best_d = torch.full((a.shape[0],), float("inf"), dtype=a.dtype)
        best_i = torch.zeros(a.shape[0], dtype=torch.int64)
        chunk = max(1, (1 << 24) // max(1, a.shape[0]))
        for s in range(0, t.shape[0], chunk):
            tc = t[s:s + chunk]
            dx = a[:, None, 0] - tc[None, :, 0]
            dy = a[:, None, 1] - tc[None, :, 1]
            dz = a[:, None, 2] - tc[None, :, 2]
            d = (dx * dx + dy * dy) + dz * dz
            dm, im = d.min(dim=1)
            # torch.min returns *a* minimal index, not necessarily the first: recover first
            first = (d == dm[:, None]).to(torch.int8).argmax(dim=1)
            upd = dm < best_d
            best_d = torch.where(upd, dm, best_d)
            best_i = torch.where(upd, first + s, best_i)
        # distances must stay differentiable in principle (never consumed by the path)
        g = p2[b][best_i]
        dd = ((p1[b] - g) ** 2)
        dists.append(((dd[:, 0] + dd[:, 1]) + dd[:, 2])[:, None])
        idxs.append(best_i[:, None])
    return _KNN(torch.stack(dists), torch.stack(idxs), None)
