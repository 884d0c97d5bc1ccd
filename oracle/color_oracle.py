"""CPU checker (test infrastructure only) for the colour-space steps of the reference's inference path
(a015_dataset.py:89 cv2.cvtColor(BGR2YCrCb) on uint8; a017_test.py:87 cv2.cvtColor(YCrCb2RGB) on float32).

PARITY UNPINNED against cv2 itself: OpenCV is not installed here and the reference holds no fixtures for this
step, so these numpy functions restate OpenCV's published arithmetic (8-bit path: 14-bit fixed point with
CV_DESCALE rounding and saturation; float path: Y + (C - 0.5) * k) and pin the HIP kernels to that restatement.
"""
import numpy as np


def bgr8_to_ycrcb8(bgr: np.ndarray) -> np.ndarray:
    b, g, r = (bgr[..., i].astype(np.int64) for i in range(3))
    desc = lambda x: (x + (1 << 13)) >> 14
    y = desc(b * 1868 + g * 9617 + r * 4899)
    cr = desc((r - y) * 11682 + (128 << 14))
    cb = desc((b - y) * 9241 + (128 << 14))
    return np.clip(np.stack([y, cr, cb], axis=-1), 0, 255).astype(np.uint8)


def ycrcb_to_rgb_f32(ycrcb: np.ndarray) -> np.ndarray:
    y, cr, cb = (ycrcb[..., i].astype(np.float32) for i in range(3))
    d = np.float32(0.5)
    b = y + (cb - d) * np.float32(1.773)
    g = y + (cb - d) * np.float32(-0.344) + (cr - d) * np.float32(-0.714)
    r = y + (cr - d) * np.float32(1.403)
    return np.stack([r, g, b], axis=-1).astype(np.float32)
