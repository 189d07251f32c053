"""Cluttered-character images with boxes, classes and colour-threshold masks.

Follows the generation rules of the reference's MNISTISEG / OMNIISEG builders
(cp_utils/create_img_from_chars.py:73-158 paste + mask rules, :161-247 image loop;
datasets/mnistiseg/mnistiseg_create.py:43-53 size classes) with numpy only: the build image
has no cv2 / imgaug and no MNIST / Omniglot glyph files, so the glyph of a class is a seeded
stroke drawing ("a black sign on a white background") instead of a dataset image.  The
layout statistics (2..5 objects, three size classes, IoU < 0.2 placement, distinct palette
colours, masks = colour window +-75 inside the box, 3x3-dilated) are the reference's; the
pixels are not.
"""
from __future__ import annotations

import numpy as np

# distinct saturated colours (the reference draws from a fixed palette, create_img_from_chars.py:40-61)
PALETTE = np.array([[230, 25, 75], [60, 180, 75], [0, 130, 200], [245, 130, 48], [145, 30, 180],
                    [70, 240, 240], [240, 50, 230], [128, 0, 0], [0, 128, 128], [128, 128, 0],
                    [0, 0, 128], [170, 110, 40]], np.uint8)
SIZES_MAX_AMOUNT = {'large': 2, 'medium': 2, 'small': 2}                 # mnistiseg_create.py:43-47
SIZES_MIN_MAX_RATIOS = {'large': (12, 15), 'medium': (8, 12), 'small': (4, 8)}   # :48-52
GLYPH = 28                                                               # MNIST-sized source glyphs


def glyph(cat_id: int, variant: int = 0) -> np.ndarray:
    """[28,28] uint8, black strokes (0) on white (255); the stroke skeleton depends on the class,
    a small jitter on the variant (stand-in for the different writers of one character)."""
    rng = np.random.RandomState(7919 * (cat_id + 1))
    n_pts = 4 + cat_id % 3
    pts = rng.uniform(4, GLYPH - 4, size=(n_pts, 2))
    pts = pts + np.random.RandomState(104729 * (cat_id + 1) + variant).uniform(-1.5, 1.5, size=pts.shape)
    img = np.full((GLYPH, GLYPH), 255, np.uint8)
    yy, xx = np.mgrid[0:GLYPH, 0:GLYPH].astype(np.float32)
    for a, b in zip(pts[:-1], pts[1:]):
        d = b - a
        t = np.clip(((yy - a[0]) * d[0] + (xx - a[1]) * d[1]) / max(float(d @ d), 1e-6), 0, 1)
        dist = np.hypot(yy - (a[0] + t * d[0]), xx - (a[1] + t * d[1]))
        img[dist <= 1.6] = 0
    return img


def cut_char(img: np.ndarray) -> np.ndarray:
    """Tight crop around the dark strokes (cut_char_img in the reference)."""
    ys, xs = np.nonzero(img < 128)
    return img[ys.min():ys.max() + 1, xs.min():xs.max() + 1]


def resize_nearest_area(img: np.ndarray, h: int, w: int) -> np.ndarray:
    """Bilinear resize of a uint8 image (cv2.resize default) with numpy."""
    H, W = img.shape
    y = (np.arange(h) + 0.5) * H / h - 0.5
    x = (np.arange(w) + 0.5) * W / w - 0.5
    y0 = np.clip(np.floor(y).astype(int), 0, H - 1); y1 = np.clip(y0 + 1, 0, H - 1)
    x0 = np.clip(np.floor(x).astype(int), 0, W - 1); x1 = np.clip(x0 + 1, 0, W - 1)
    wy = np.clip(y - y0, 0, 1)[:, None]; wx = np.clip(x - x0, 0, 1)[None, :]
    f = img.astype(np.float32)
    out = (f[y0][:, x0] * (1 - wy) * (1 - wx) + f[y0][:, x1] * (1 - wy) * wx +
           f[y1][:, x0] * wy * (1 - wx) + f[y1][:, x1] * wy * wx)
    return np.clip(np.rint(out), 0, 255).astype(np.uint8)


def _iou_one_to_many(box, boxes):
    y0 = np.maximum(box[0], boxes[:, 0]); x0 = np.maximum(box[1], boxes[:, 1])
    y1 = np.minimum(box[2], boxes[:, 2]); x1 = np.minimum(box[3], boxes[:, 3])
    inter = np.clip(y1 - y0, 0, None) * np.clip(x1 - x0, 0, None)
    area = lambda b: (b[..., 2] - b[..., 0]) * (b[..., 3] - b[..., 1])
    return inter / (area(box) + area(boxes) - inter)


def paste_colored_char(img, char, bboxes, colors, rng, iou_max=0.2):
    """create_img_from_chars.py:73-133: up to 50 placement attempts with IoU < iou_max against the
    boxes placed so far, a palette colour not used in this image, paste where the tinted glyph is
    darker than 245.  Returns False when no place was found."""
    size = img.shape[0]
    h, w = char.shape
    if h >= size or w >= size:
        return False
    for _ in range(50):
        y0, x0 = rng.randint(0, size - h), rng.randint(0, size - w)
        box = np.array([y0, x0, y0 + h, x0 + w])
        if len(bboxes) == 0 or _iou_one_to_many(box, np.asarray(bboxes)).max() < iou_max:
            break
    else:
        return False
    used = {tuple(c) for c in colors}
    free = [i for i, c in enumerate(PALETTE) if tuple(c) not in used]
    color = PALETTE[rng.choice(free)]
    tinted = 255.0 - (255 - char).astype(np.float32)[..., None] * (1 - color.astype(np.float32) / 255)
    tinted = tinted.astype(np.uint8)
    sel = (tinted < 245).any(-1)
    img[y0:y0 + h, x0:x0 + w][sel] = tinted[sel]
    bboxes.append(box)
    colors.append(color)
    return True


def char_mask_by_color(img, box, color, shift=75) -> np.ndarray:
    """create_img_from_chars.py:136-158: pixels of the box within +-shift of the colour, dilated 3x3."""
    y0, x0, y1, x1 = box
    roi = img[y0:y1, x0:x1].astype(np.int32)
    c = color.astype(np.int32)
    m = ((roi >= np.maximum(c - shift, 0)) & (roi <= np.minimum(c + shift, 255))).all(-1)
    p = np.pad(m, 1)
    d = np.zeros_like(m)
    for dy in range(3):
        for dx in range(3):
            d |= p[dy:dy + m.shape[0], dx:dx + m.shape[1]]
    out = np.zeros(img.shape[:2], bool)
    out[y0:y1, x0:x1] = d
    return out


def make_image(seed: int, size: int, cats) -> dict:
    """One cluttered image: 2..5 characters of the given classes (create_ds loop, :183-225)."""
    rng = np.random.RandomState(seed)
    while True:
        img = np.full((size, size, 3), 255, np.uint8)
        bboxes, colors, cat_ids = [], [], []
        for name in sorted(SIZES_MAX_AMOUNT):
            for _ in range(rng.randint(0, SIZES_MAX_AMOUNT[name] + 1)):
                cat = int(cats[rng.randint(0, len(cats))])
                ch = cut_char(glyph(cat, int(rng.randint(0, 1000))))
                ratio = rng.uniform(*SIZES_MIN_MAX_RATIOS[name]) * size / 512.0
                ch = resize_nearest_area(ch, max(int(ch.shape[0] * ratio), 2), max(int(ch.shape[1] * ratio), 2))
                if paste_colored_char(img, ch, bboxes, colors, rng):
                    cat_ids.append(cat)
            if len(bboxes) > 4:
                break
        if len(bboxes) >= 2:
            break
    masks = np.stack([char_mask_by_color(img, b, c) for b, c in zip(bboxes, colors)])
    return dict(img=img, bboxes=np.asarray(bboxes, np.float32), cat_ids=np.asarray(cat_ids, np.int64),
                isegmaps=masks)


def crop_support(img, box, mask, out_size: int, fill_ratio: float = 0.8):
    """Support crop: the instance centred in a square window whose side is max(h, w) / fill_ratio
    (base_fst.py:264-265 offset rule), white padding outside the image, resized to out_size^2.
    Returns (crop uint8 [S,S,3], box YXYX float32 in crop pixels, mask bool [S,S])."""
    y0, x0, y1, x1 = [float(v) for v in box]
    side = max(y1 - y0, x1 - x0) / fill_ratio
    cy, cx = (y0 + y1) / 2, (x0 + x1) / 2
    wy0, wx0 = cy - side / 2, cx - side / 2
    pos = (np.arange(out_size) + 0.5) * side / out_size
    ys = np.floor(wy0 + pos).astype(int); xs = np.floor(wx0 + pos).astype(int)
    ok_y = (ys >= 0) & (ys < img.shape[0]); ok_x = (xs >= 0) & (xs < img.shape[1])
    crop = np.full((out_size, out_size, 3), 255, np.uint8)
    m = np.zeros((out_size, out_size), bool)
    yy, xx = np.clip(ys, 0, img.shape[0] - 1), np.clip(xs, 0, img.shape[1] - 1)
    inside = ok_y[:, None] & ok_x[None, :]
    crop[inside] = img[yy][:, xx][inside]
    m[inside] = mask[yy][:, xx][inside]
    s = out_size / side
    nb = np.array([(y0 - wy0) * s, (x0 - wx0) * s, (y1 - wy0) * s, (x1 - wx0) * s], np.float32)
    return crop, nb, m
