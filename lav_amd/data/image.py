"""The three OpenCV calls the reference's data loaders make, restated on numpy + PIL (this image has no cv2):
`cv2.imdecode`, `cv2.getRotationMatrix2D` and `cv2.warpAffine(..., flags=cv2.INTER_LINEAR)` on 8-bit images
(lav/utils/datasets/basic_dataset.py:84-101, bev_dataset.py:81-85).

warp_affine_linear follows OpenCV 4's fixed-point path for 8-bit bilinear warps (imgproc/src/imgwarp.cpp: warpAffine ->
remapBilinear): the inverse map is evaluated in 1/1024 pixel integers with a rounding offset of 16, source coordinates are
cut to 1/32 pixel, the four weights are (32 - a)(32 - b) * 32 / 32768 exactly, the sum is rounded with +2^14 >> 15, and
pixels outside the source read the constant border 0.  PARITY UNPINNED against OpenCV itself (none here to compare with);
the loaders only use the result through `> 0`, so only pixels whose interpolated value rounds to zero could differ.
"""
from __future__ import annotations

import io

import numpy as np

IMREAD_GRAYSCALE, IMREAD_COLOR = 0, 1
INTER_LINEAR = 1


def imdecode(buf, mode: int) -> np.ndarray:
    """Encoded image bytes -> (H, W) uint8 for IMREAD_GRAYSCALE, (H, W, 3) BGR uint8 for IMREAD_COLOR."""
    from PIL import Image
    im = Image.open(io.BytesIO(bytes(buf)))
    if mode == IMREAD_GRAYSCALE:
        return np.asarray(im.convert("L"))
    if mode == IMREAD_COLOR:
        return np.ascontiguousarray(np.asarray(im.convert("RGB"))[..., ::-1])
    raise NotImplementedError(f"imdecode mode {mode}")


def imencode_png(img: np.ndarray) -> bytes:
    """(H, W) uint8 -> PNG bytes (what the data collector stores under map_*/sem_* keys); for building test routes."""
    from PIL import Image
    out = io.BytesIO()
    Image.fromarray(np.ascontiguousarray(img)).save(out, format="PNG")
    return out.getvalue()


def rotation_matrix_2d(center, angle_deg: float, scale: float = 1.0) -> np.ndarray:
    """cv2.getRotationMatrix2D: positive angles rotate counter-clockwise about `center` (x, y), origin top-left."""
    cx, cy = float(np.float32(center[0])), float(np.float32(center[1]))      # cv::Point2f
    a = angle_deg * np.pi / 180.0
    alpha, beta = np.cos(a) * scale, np.sin(a) * scale
    return np.array([[alpha, beta, (1 - alpha) * cx - beta * cy], [-beta, alpha, beta * cx + (1 - alpha) * cy]], np.float64)


def warp_affine_linear(image: np.ndarray, M: np.ndarray) -> np.ndarray:
    """cv2.warpAffine(image, M, (W, H), flags=INTER_LINEAR) for uint8 images of shape (H, W) or (H, W, C); constant border 0."""
    if image.dtype != np.uint8:
        raise TypeError("warp_affine_linear restates the 8-bit fixed-point path only")
    src = image if image.ndim == 3 else image[..., None]
    H, W, _ = src.shape
    m = np.array(M, np.float64).reshape(2, 3)
    # forward map -> inverse map, exactly as warpAffine does it
    D = m[0, 0] * m[1, 1] - m[0, 1] * m[1, 0]
    D = 1.0 / D if D != 0 else 0.0
    a11, a22 = m[1, 1] * D, m[0, 0] * D
    i00, i01, i10, i11 = a11, m[0, 1] * -D, m[1, 0] * -D, a22
    b1 = -i00 * m[0, 2] - i01 * m[1, 2]
    b2 = -i10 * m[0, 2] - i11 * m[1, 2]
    AB_SCALE, ROUND = 1024, 16
    xs = np.arange(W, dtype=np.float64)
    ys = np.arange(H, dtype=np.float64)
    adelta = np.rint(i00 * xs * AB_SCALE).astype(np.int64)
    bdelta = np.rint(i10 * xs * AB_SCALE).astype(np.int64)
    X0 = np.rint((i01 * ys + b1) * AB_SCALE).astype(np.int64) + ROUND
    Y0 = np.rint((i11 * ys + b2) * AB_SCALE).astype(np.int64) + ROUND
    X = (X0[:, None] + adelta[None, :]) >> 5
    Y = (Y0[:, None] + bdelta[None, :]) >> 5
    sx = np.clip(X >> 5, -32768, 32767)
    sy = np.clip(Y >> 5, -32768, 32767)
    fx, fy = (X & 31), (Y & 31)
    w = np.stack([(32 - fy) * (32 - fx), (32 - fy) * fx, fy * (32 - fx), fy * fx], 0) * 32          # sums to 32768

    def tap(yy, xx):
        ok = (yy >= 0) & (yy < H) & (xx >= 0) & (xx < W)
        v = src[np.clip(yy, 0, H - 1), np.clip(xx, 0, W - 1)].astype(np.int64)
        return v * ok[..., None]

    acc = (tap(sy, sx) * w[0][..., None] + tap(sy, sx + 1) * w[1][..., None]
           + tap(sy + 1, sx) * w[2][..., None] + tap(sy + 1, sx + 1) * w[3][..., None])
    out = ((acc + (1 << 14)) >> 15).astype(np.uint8)
    return out if image.ndim == 3 else out[..., 0]
