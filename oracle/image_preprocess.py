"""CPU restatement of the reference's image pre-processing -- TEST INFRASTRUCTURE ONLY.

starvector/data/util.py:40-68 (`ImageTrainProcessor`): RGBA -> composite on white (PIL `paste` with the alpha band as
mask), white pad to square, `transforms.Resize(size, BICUBIC)` on the PIL image (= Pillow's two-pass antialiased
resampler, 8 bits per channel, fixed point), `ToTensor` (/255), `Normalize(mean, std)`.

Pillow is the un-vendored dependency that holds the arithmetic (libImaging/Resample.c, libImaging/Paste.c); this file
restates it in numpy integer arithmetic and `pin()` checks it bit for bit against Pillow itself (which is installed in the
build container and on the GPU box).  The HIP kernels (star-vector_amd/csrc/preprocess.hip) are tested against it and,
directly, against Pillow + torch.
"""
from __future__ import annotations

import math

import numpy as np

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)      # data/util.py:34-37
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)
PRECISION_BITS = 32 - 8 - 2                            # Resample.c: fixed-point coefficients for 8-bit channels


def _bicubic(x: float) -> float:
    """Resample.c bicubic_filter, a = -0.5."""
    a = -0.5
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def resample_coeffs(in_size: int, out_size: int):
    """Resample.c precompute_coeffs + normalize_coeffs_8bpc for box (0, in_size): per output index the first input
    index, the tap count and the int32 taps (sum == 1 << PRECISION_BITS up to rounding).  Double precision, like C."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    taps = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        xmin = max(xmin, 0)
        xmax = int(center + support + 0.5)
        xmax = min(xmax, in_size) - xmin
        k = [_bicubic((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = sum(k)                                    # C accumulates in the same left-to-right order
        acc = 0.0
        for v in k:
            acc += v
        ww = acc
        for x in range(xmax):
            v = k[x] / ww if ww != 0.0 else k[x]
            taps[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, taps


def _clip8(v: np.ndarray) -> np.ndarray:
    return np.clip(v >> PRECISION_BITS, 0, 255).astype(np.uint8)


def resize_bicubic_u8(img: np.ndarray, out_size: int) -> np.ndarray:
    """Pillow `Image.resize((out, out), BICUBIC)` on a uint8 [H, W, C] image: horizontal pass to uint8, then
    vertical pass (ImagingResampleHorizontal_8bpc / ImagingResampleVertical_8bpc).  A pass whose input size already
    equals the output size has identity taps (Pillow skips it)."""
    h, w, _ = img.shape
    bh, th = resample_coeffs(w, out_size)
    bv, tv = resample_coeffs(h, out_size)
    src = img.astype(np.int64)
    tmp = np.empty((h, out_size, img.shape[2]), dtype=np.uint8)
    for xo in range(out_size):
        x0, n = bh[xo]
        acc = (src[:, x0:x0 + n, :] * th[xo, :n].astype(np.int64)[None, :, None]).sum(axis=1) + (1 << (PRECISION_BITS - 1))
        tmp[:, xo, :] = _clip8(acc)
    out = np.empty((out_size, out_size, img.shape[2]), dtype=np.uint8)
    t64 = tmp.astype(np.int64)
    for yo in range(out_size):
        y0, n = bv[yo]
        acc = (t64[y0:y0 + n, :, :] * tv[yo, :n].astype(np.int64)[:, None, None]).sum(axis=0) + (1 << (PRECISION_BITS - 1))
        out[yo] = _clip8(acc)
    return out


def composite_on_white(rgba: np.ndarray) -> np.ndarray:
    """`background.paste(img, mask=alpha)` on a white RGB image (Paste.c paste_mask_L): per channel
    BLEND8(mask, 255, in) = DIV255(255 * (255 - mask) + in * mask), DIV255(a) = ((t >> 8) + t) >> 8 with t = a + 128."""
    a = rgba[..., 3:4].astype(np.int64)
    c = rgba[..., :3].astype(np.int64)
    t = 255 * (255 - a) + c * a + 128
    return (((t >> 8) + t) >> 8).astype(np.uint8)


def pad_to_square_white(rgb: np.ndarray) -> np.ndarray:
    """data/util.py:56-62."""
    h, w, _ = rgb.shape
    m = max(h, w)
    out = np.full((m, m, 3), 255, dtype=np.uint8)
    left, top = (m - w) // 2, (m - h) // 2
    out[top:top + h, left:left + w] = rgb
    return out


def preprocess(pixels: np.ndarray, size: int = 224, mean=CLIP_MEAN, std=CLIP_STD) -> np.ndarray:
    """uint8 [H, W, 3 | 4] -> float32 [3, size, size], the tensor `ImageTrainProcessor.__call__` returns."""
    rgb = composite_on_white(pixels) if pixels.shape[2] == 4 else pixels
    sq = pad_to_square_white(rgb)
    if sq.shape[0] != size:
        sq = resize_bicubic_u8(sq, size)
    x = sq.astype(np.float32).transpose(2, 0, 1) / np.float32(255.0)                       # ToTensor
    m = np.asarray(mean, dtype=np.float32).reshape(3, 1, 1)
    s = np.asarray(std, dtype=np.float32).reshape(3, 1, 1)
    return ((x - m) / s).astype(np.float32)                                                # Normalize


def preprocess_siglip(pixels: np.ndarray, size: int = 384, mean=(0.5, 0.5, 0.5), std=(0.5, 0.5, 0.5)) -> np.ndarray:
    """HF SiglipImageProcessor (the `AutoProcessor` of the v2 tower, image_encoder.py:45-48,116-117): convert("RGB") drops
    alpha, the image is STRETCHED to size x size (same resampler), rescale = float32(float64(u) * (1/255)), normalise."""
    rgb = np.ascontiguousarray(pixels[..., :3])
    if rgb.shape[0] != size or rgb.shape[1] != size:
        rgb = resize_bicubic_u8(rgb, size)
    x = (rgb.astype(np.float64) * (1 / 255)).astype(np.float32).transpose(2, 0, 1)
    m = np.asarray(mean, dtype=np.float32).reshape(3, 1, 1)
    s = np.asarray(std, dtype=np.float32).reshape(3, 1, 1)
    return ((x - m) / s).astype(np.float32)


def pin_siglip(verbose: bool = True) -> None:
    """The SigLIP recipe against HF's own PIL image processor (when transformers is importable) and Pillow."""
    from PIL import Image
    rng = np.random.default_rng(1)
    try:
        from transformers.models.siglip.image_processing_pil_siglip import SiglipImageProcessorPil as HFP
    except Exception:                                      # older transformers: the PIL processor is the default class
        from transformers import SiglipImageProcessor as HFP
    ip = HFP(size={"height": 384, "width": 384}, resample=3, do_rescale=True, rescale_factor=1 / 255, do_normalize=True,
             image_mean=[0.5, 0.5, 0.5], image_std=[0.5, 0.5, 0.5], do_convert_rgb=True)
    for (w, h, c) in [(517, 300, 4), (384, 384, 3), (224, 224, 3), (1000, 120, 3), (384, 100, 4)]:
        px = rng.integers(0, 256, size=(h, w, c), dtype=np.uint8)
        ref = ip(images=[Image.fromarray(px, "RGBA" if c == 4 else "RGB")], return_tensors="np")["pixel_values"][0]
        assert np.array_equal(preprocess_siglip(px).view(np.int32), np.asarray(ref, dtype=np.float32).view(np.int32)), (w, h, c)
        if verbose:
            print(f"[image_preprocess/siglip] {w}x{h}x{c}: == HF SiglipImageProcessor (PIL), bit for bit")


def pin(verbose: bool = True) -> None:
    """Check the restatement bit for bit against Pillow (+ the float steps against torch) on random images."""
    import torch
    from PIL import Image
    rng = np.random.default_rng(0)
    cases = [(224, 224, 3), (300, 300, 3), (100, 60, 4), (517, 333, 3), (64, 200, 4), (1024, 768, 3), (50, 50, 3),
             (223, 225, 4), (448, 448, 4)]
    for (w, h, c) in cases:
        px = rng.integers(0, 256, size=(h, w, c), dtype=np.uint8)
        if c == 4:
            px[..., 3] = rng.choice([0, 255, 128, 7], size=(h, w), p=[0.3, 0.4, 0.2, 0.1])
        img = Image.fromarray(px, "RGBA" if c == 4 else "RGB")
        if c == 4:
            bg = Image.new("RGB", img.size, (255, 255, 255))
            bg.paste(img, mask=img.split()[3])
            assert np.array_equal(np.asarray(bg), composite_on_white(px)), ("composite", w, h)
            img = bg
        m = max(w, h)
        canvas = Image.new("RGB", (m, m), (255, 255, 255))
        canvas.paste(img, ((m - w) // 2, (m - h) // 2))
        if m != 224:
            canvas = canvas.resize((224, 224), Image.BICUBIC)
        ref_u8 = np.asarray(canvas)
        mine = pad_to_square_white(composite_on_white(px) if c == 4 else px)
        if m != 224:
            mine = resize_bicubic_u8(mine, 224)
        assert np.array_equal(mine, ref_u8), ("resize", w, h, int(np.abs(mine.astype(int) - ref_u8.astype(int)).max()))
        t = torch.from_numpy(ref_u8.copy()).permute(2, 0, 1).float().div(255.0)
        t = (t - torch.tensor(CLIP_MEAN).view(3, 1, 1)) / torch.tensor(CLIP_STD).view(3, 1, 1)
        assert np.array_equal(preprocess(px).view(np.int32), t.numpy().view(np.int32)), ("normalize", w, h)
        if verbose:
            print(f"[image_preprocess] {w}x{h}x{c}: composite / pad / bicubic resize / normalize == Pillow + torch, bit for bit")


if __name__ == "__main__":
    pin()
    pin_siglip()
