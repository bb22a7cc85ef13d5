"""Time-stamp association for TUM RGB-D sequences (reference datasets/tumutils.py:95-216, itself derived from
the TUM benchmark tools).  Restated from the behaviour: same function names, arguments and results.

The Kinect delivers colour and depth unsynchronised, so `rgb.txt`, `depth.txt` and `groundtruth.txt` carry
different time stamps that have to be paired by proximity."""
import sys
import warnings
from typing import Optional

import numpy as np

__all__ = ["read_trajectory", "read_file_list", "associate", "transform44"]

_EPS = np.finfo(float).eps * 4.0


def _fields(line: str):
    return [v for v in line.replace(",", " ").replace("\t", " ").split(" ") if v.strip() != ""]


def transform44(l: tuple):
    """(stamp, tx, ty, tz, qx, qy, qz, qw) -> 4x4 float64 (reference :57-92)."""
    t = l[1:4]
    q = np.array(l[4:8], dtype=np.float64)
    nq = float(np.dot(q, q))
    T = np.eye(4, dtype=np.float64)
    T[:3, 3] = t
    if nq < _EPS:
        return T
    q = q * np.sqrt(2.0 / nq)
    o = np.outer(q, q)
    T[:3, :3] = [[1.0 - o[1, 1] - o[2, 2], o[0, 1] - o[2, 3], o[0, 2] + o[1, 3]],
                 [o[0, 1] + o[2, 3], 1.0 - o[0, 0] - o[2, 2], o[1, 2] - o[0, 3]],
                 [o[0, 2] - o[1, 3], o[1, 2] + o[0, 3], 1.0 - o[0, 0] - o[1, 1]]]
    return T


def read_trajectory(filename: str, matrix: bool = True):
    """{stamp (str): 4x4 pose or (tx,ty,tz,qx,qy,qz,qw)} from a `groundtruth.txt`-style file; comment lines,
    all-zero quaternions and lines with NaNs are skipped (reference :95-143)."""
    traj = {}
    with open(filename) as f:
        for i, line in enumerate(f.read().split("\n")):
            if not line or line[0] == "#":
                continue
            vals = _fields(line)
            if not vals:
                continue
            row = [vals[0]] + [float(v) for v in vals[1:]]
            if row[4:8] == [0, 0, 0, 0]:
                continue
            if any(np.isnan(v) for v in row[1:]):
                sys.stderr.write("Warning: line %d of file '%s' has NaNs, skipping line\n" % (i, filename))
                continue
            traj[row[0]] = transform44(row) if matrix else row[1:8]
    return traj


def read_file_list(filename: str, start: Optional[int] = None, end: Optional[int] = None):
    """{stamp (str): [d1, d2, ...]} of the data lines [start, end) of a "stamp d1 d2 ..." text file
    (reference :146-179)."""
    with open(filename) as f:
        lines = f.read().split("\n")
    rows = [_fields(line) for line in lines if len(line) > 0 and line[0] != "#"]
    start = 0 if start is None else start
    if end is None:
        end = len(lines)
    if end > len(lines):
        warnings.warn('"end" was larger than number of frames in "{0}": {1} > {2}'.format(filename, end, len(lines)))
    return {r[0]: r[1:] for r in rows[start:end] if len(r) > 1}


def associate(first_dict: dict, second_dict: dict, offset: float, max_difference: float):
    """Pairs (stamp1, stamp2) with |stamp1 - (stamp2 + offset)| < max_difference, closest pairs first, every
    stamp used at most once; returned sorted (reference :182-216).  The candidate pairs of each stamp are found
    by bisection in the sorted second list, so the cost is O(n log n + matches) instead of all pairs."""
    import bisect

    second = sorted((float(b) + offset, b) for b in second_dict.keys())
    keys2 = [s for s, _ in second]
    candidates = []
    for a in first_dict.keys():
        fa = float(a)
        lo = bisect.bisect_left(keys2, fa - max_difference)
        hi = bisect.bisect_right(keys2, fa + max_difference)
        for s, b in second[max(lo - 1, 0): hi + 1]:
            diff = abs(float(a) - (float(b) + offset))
            if diff < max_difference:
                candidates.append((diff, a, b))
    candidates.sort()
    free1, free2 = set(first_dict.keys()), set(second_dict.keys())
    matches = []
    for _, a, b in candidates:
        if a in free1 and b in free2:
            free1.discard(a)
            free2.discard(b)
            matches.append((a, b))
    matches.sort()
    return matches
