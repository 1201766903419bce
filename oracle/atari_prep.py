"""CPU restatement of border-atari-env's frame preprocessing (test infrastructure; never imported by the product).

Follows border-atari-env/src/env.rs:
  skip_and_max            :126-157  element-wise max of the last two RGB frames of a skip-4 step
  warp_and_grayscale      :171-195  image::imageops::resize(&img, 84, 84, Triangle), then per pixel
                                    ((0.299 * c2 as f32) + (0.587 * c1 as f32) + (0.114 * c0 as f32)) as u8
                                    (the reference names the three channels (b, g, r) although the emulator renders RGB:
                                    channel 0 gets 0.114 - kept as is)
  stack_frame             :197-209  frames[1..4] <- frames[0..3]; frames[0] <- new frame
  reset                   :263-296  all four slots <- warp(first frame)
  clip_reward             :159-169  train: sign(r) (0 stays 0); eval: r

The resize lives in a third-party dependency absent from /root/reference: `image = "0.23.14"` (Cargo.toml:50).  Its
`imageops::resize` = `vertical_sample` (height) followed by `horizontal_sample` (width), both restated below from the
crate's published source (src/imageops/sample.rs) as remembered:
  * Triangle: kernel(x) = 1 - |x| for |x| < 1 else 0, support 1.0;
  * ratio = in / out (f32), sratio = max(ratio, 1), src_support = support * sratio;
  * for each output coordinate o: centre = (o + 0.5) * ratio; left = clamp(floor(centre - src_support), 0, in - 1);
    right = clamp(ceil(centre + src_support), left + 1, in); centre -= 0.5;
    w_i = kernel((i - centre) / sratio) for i in left..right, sum accumulated in that order;
  * each channel: t = sum_i pixel_i * w_i (f32, in order, no fused multiply-add), t / sum, clamp to [0, 255], round half
    away from zero (`FloatNearest`), store as u8 - the intermediate image between the two passes is u8 in 0.23.
PARITY UNPINNED: no test vector of the reference or of the image crate exists offline, so this restatement is not checked
against an output of the real crate; the device kernel is checked bit-exactly against this file.
"""
from __future__ import annotations

import numpy as np

F = np.float32


def _weights(n_in: int, n_out: int):
    """Per output coordinate: (left, [w_i ...], sum) exactly as sample.rs computes them (f32 throughout)."""
    ratio = F(n_in) / F(n_out)
    sratio = F(1.0) if ratio < F(1.0) else ratio
    src_support = F(1.0) * sratio
    out = []
    for o in range(n_out):
        centre = (F(o) + F(0.5)) * ratio
        left = int(np.floor(centre - src_support))
        left = min(max(left, 0), n_in - 1)
        right = int(np.ceil(centre + src_support))
        right = min(max(right, left + 1), n_in)
        centre = centre - F(0.5)
        ws, total = [], F(0.0)
        for i in range(left, right):
            x = (F(i) - centre) / sratio
            ax = np.abs(x)
            w = F(1.0) - ax if ax < F(1.0) else F(0.0)
            ws.append(F(w))
            total = F(total + w)
        out.append((left, ws, total))
    return out


def _sample_axis0(img: np.ndarray, n_out: int) -> np.ndarray:
    """Resample axis 0 of a u8 array [n_in, ...] to n_out rows (one pass of sample.rs)."""
    n_in = img.shape[0]
    res = np.empty((n_out,) + img.shape[1:], np.uint8)
    for o, (left, ws, total) in enumerate(_weights(n_in, n_out)):
        t = np.zeros(img.shape[1:], F)
        for k, w in enumerate(ws):
            t = (t + img[left + k].astype(F) * w).astype(F)      # mul, then add: two roundings
        t = (t / total).astype(F)
        t = np.minimum(np.maximum(t, F(0.0)), F(255.0))
        res[o] = _round_half_away(t)      # f32::round: half away from zero (values are >= 0 here)
    return res


def _round_half_away(t: np.ndarray) -> np.ndarray:
    fl = np.floor(t)
    frac = t - fl                      # exact in f32 for 0 <= t <= 255
    return (fl + (frac >= F(0.5))).astype(np.uint8)


def resize_triangle(img: np.ndarray, width: int = 84, height: int = 84) -> np.ndarray:
    """image::imageops::resize(&img, width, height, Triangle) for a u8 image [H][W][C]."""
    tmp = _sample_axis0(img, height)                                   # vertical_sample
    return np.ascontiguousarray(_sample_axis0(tmp.transpose(1, 0, 2), width).transpose(1, 0, 2))   # horizontal_sample


def grayscale(img: np.ndarray) -> np.ndarray:
    """env.rs:178-187, channel order as in the reference; `as u8` truncates (and saturates)."""
    c0, c1, c2 = (img[..., k].astype(F) for k in range(3))
    g = ((F(0.299) * c2).astype(F) + (F(0.587) * c1).astype(F)).astype(F)
    g = (g + (F(0.114) * c0).astype(F)).astype(F)
    return np.minimum(np.floor(g), F(255.0)).astype(np.uint8)


def warp_and_grayscale(frame_rgb: np.ndarray) -> np.ndarray:
    """[H][W][3] u8 -> [84][84] u8."""
    return grayscale(resize_triangle(frame_rgb, 84, 84))


def clip_reward(r: float, train: bool) -> float:
    if not train:
        return float(r)
    return 0.0 if r == 0.0 else float(np.sign(r))


class FrameStack:
    """The `frames: vec![0; 4*84*84]` state of one environment (newest frame first)."""

    def __init__(self):
        self.frames = np.zeros((4, 84, 84), np.uint8)

    def reset(self, frame_rgb: np.ndarray) -> np.ndarray:
        self.frames[:] = warp_and_grayscale(frame_rgb)[None]
        return self.frames.copy()

    def step(self, frame_a: np.ndarray, frame_b: np.ndarray) -> np.ndarray:
        g = warp_and_grayscale(np.maximum(frame_a, frame_b))
        self.frames[1:] = self.frames[:3].copy()
        self.frames[0] = g
        return self.frames.copy()
