"""list <-> zero-padded conversions for ragged batches (the job of reference
structures/structutils.py:47-129; written against the same contract, not its text)."""
from typing import List, Optional, Sequence

import torch

__all__ = ["list_to_padded", "padded_to_list", "numpy_to_plotly_image", "img_to_b64str"]


def list_to_padded(x: List[torch.Tensor], pad_size: Optional[Sequence[int]] = None, pad_value: float = 0.0,
                   equisized: bool = False) -> torch.Tensor:
    """B tensors (N_b, C_b) -> one (B, pad_size[0], pad_size[1]) tensor filled with ``pad_value``."""
    if equisized:
        return torch.stack(x, 0)
    if pad_size is None:
        nonempty = [y for y in x if len(y) > 0]
        rows = max(y.shape[0] for y in nonempty)
        cols = max(y.shape[1] for y in nonempty)
    else:
        if len(pad_size) != 2:
            raise ValueError("Pad size must contain target size for 1st and 2nd dim")
        rows, cols = pad_size
    out = torch.full((len(x), int(rows), int(cols)), pad_value, dtype=x[0].dtype, device=x[0].device)
    for b, y in enumerate(x):
        if len(y) == 0:
            continue
        if y.ndim != 2:
            raise ValueError("Supports only 2-dimensional tensor items")
        out[b, : y.shape[0], : y.shape[1]] = y
    return out


def padded_to_list(x: torch.Tensor, split_size=None) -> List[torch.Tensor]:
    """(B, N, C) -> list of B tensors, optionally cropped to split_size[b] (int or (rows, cols))."""
    if x.ndim != 3:
        raise ValueError("Supports only 3-dimensional input tensors")
    items = list(x.unbind(0))
    if split_size is None:
        return items
    if len(split_size) != x.shape[0]:
        raise ValueError("Split size must be of same length as inputs first dimension")
    for b, s in enumerate(split_size):
        if isinstance(s, int):
            items[b] = items[b][:s]
        elif len(s) == 2:
            items[b] = items[b][: s[0], : s[1]]
        else:
            raise ValueError("Support only for 2-dimensional unbinded tensor. Split size for more dimensions provided")
    return items


# ---------------------------------------------------------------------- viewer helpers (reference :127-178)
def img_to_b64str(img, quality: int = 95) -> str:
    """uint8 (H, W) or (H, W, 3) array -> "data:image/jpeg;base64,..." (JPEG through PIL; the reference uses cv2)."""
    import base64
    import io

    import numpy as np
    from PIL import Image

    if not isinstance(img, np.ndarray):
        raise TypeError(f"img must be of type np.ndarray, but was {type(img)}")
    if img.ndim != 2 and img.ndim != 3:
        raise ValueError(f"img.ndim must be 2 or 3, but was {img.ndim}")
    buf = io.BytesIO()
    Image.fromarray(img).save(buf, format="JPEG", quality=quality)
    return "data:image/jpeg;base64," + base64.b64encode(buf.getvalue()).decode("utf-8")


def numpy_to_plotly_image(img, name=None, is_depth: bool = False, scale=None, quality: int = 95):
    """`plotly.graph_objects.Image` of an RGB or depth array with a hover label (colour / depth, optional scale)."""
    import plotly.graph_objects as go

    what, value = ("depth", "%{z[0]}") if is_depth else ("color", "[%{z[0]}, %{z[1]}, %{z[2]}]")
    hover = "x: %{x}<br>y: %{y}<br>" + what + ": " + value
    if scale is not None:
        scale = int(scale) if int(scale) == scale else scale
        hover += f"<br>scale: x{scale}<br>"
    return go.Image(source=img_to_b64str(img, quality), hovertemplate=hover + "<extra></extra>", name=name)
