"""Stand-in for chamferdist==1.0.0 `knn_points` (K=1 only): brute force, squared L2,
x->y->z accumulation, first (lowest-index) minimum wins.  Our own code -- see README.md."""
from collections import namedtuple

import torch

_KNN = namedtuple("KNN", "dists idx knn")


def knn_points(p1, p2, lengths1=None, lengths2=None, K=1, version=-1,
               return_nn=False, return_sorted=True):
    assert K == 1 and p1.ndim == 3 and p2.ndim == 3
    dists, idxs = [], []
    for b in range(p1.shape[0]):
        a, t = p1[b].detach(), p2[b].detach()
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
