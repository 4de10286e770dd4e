"""Evaluator with the reference's surface (evaluators/gdb_nerf.py:12-151): `evaluate(output, batch)`
accumulates masked PSNR, SSIM and optional depth errors; `summarize()` prints per-scene rows and
returns {'psnr', 'ssim'[, 'lpips']}.

skimage, lpips and cv2 are not in the MI355X image, so the two skimage metrics are restated in numpy:
  * peak_signal_noise_ratio(gt[mask], pred[mask], data_range=1)           (reference :82)
  * structural_similarity(gt, pred, channel_axis=-1) with skimage's defaults for float images:
    7x7 uniform window, K1 = 0.01, K2 = 0.03, sample covariance, data_range 2 (dtype range of
    float images), mean over the interior and the channels                  (reference :86)
LPIPS needs the `lpips` package and its VGG weights; `eval_lpips: True` without them is an error."""
import math
import os
import struct
import zlib
from collections import defaultdict

import numpy as np
from scipy.ndimage import uniform_filter


def psnr(gt: np.ndarray, pred: np.ndarray, data_range: float = 1.0) -> float:
    mse = float(np.mean((np.asarray(gt, np.float64) - np.asarray(pred, np.float64)) ** 2))
    return float("inf") if mse == 0 else 10.0 * math.log10(data_range ** 2 / mse)


def ssim(gt: np.ndarray, pred: np.ndarray, win: int = 7, data_range: float = 2.0) -> float:
    """(H,W,C) float images."""
    c1, c2 = (0.01 * data_range) ** 2, (0.03 * data_range) ** 2
    norm = win * win / (win * win - 1.0)
    pad = (win - 1) // 2
    vals = []
    for ch in range(gt.shape[-1]):
        x, y = gt[..., ch].astype(np.float64), pred[..., ch].astype(np.float64)
        ux, uy = uniform_filter(x, win), uniform_filter(y, win)
        vx = norm * (uniform_filter(x * x, win) - ux * ux)
        vy = norm * (uniform_filter(y * y, win) - uy * uy)
        vxy = norm * (uniform_filter(x * y, win) - ux * uy)
        s = ((2 * ux * uy + c1) * (2 * vxy + c2)) / ((ux * ux + uy * uy + c1) * (vx + vy + c2))
        vals.append(s[pad:-pad, pad:-pad].mean())
    return float(np.mean(vals))


def _resize_bilinear(img: np.ndarray, size_hw) -> np.ndarray:
    """cv2.resize(..., INTER_LINEAR) for a single-channel map (half-pixel centres, edge clamp)."""
    H, W = img.shape
    h, w = size_hw
    ys = np.clip((np.arange(h) + 0.5) * H / h - 0.5, 0, H - 1)
    xs = np.clip((np.arange(w) + 0.5) * W / w - 0.5, 0, W - 1)
    y0, x0 = np.floor(ys).astype(int), np.floor(xs).astype(int)
    y1, x1 = np.minimum(y0 + 1, H - 1), np.minimum(x0 + 1, W - 1)
    fy, fx = (ys - y0)[:, None], (xs - x0)[None, :]
    top = img[y0][:, x0] * (1 - fx) + img[y0][:, x1] * fx
    bot = img[y1][:, x0] * (1 - fx) + img[y1][:, x1] * fx
    return top * (1 - fy) + bot * fy


def write_png(path: str, rgb_u8: np.ndarray) -> None:
    """Minimal 8-bit RGB PNG writer (cv2.imwrite stand-in)."""
    h, w, _ = rgb_u8.shape
    raw = b"".join(b"\x00" + rgb_u8[y].tobytes() for y in range(h))
    chunk = lambda tag, data: struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) +
                chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


class Evaluator:
    def __init__(self, cfg):
        self.cfg = cfg
        self._reset()
        self.loss_fn_vgg = None
        if getattr(cfg, "eval_lpips", False):
            try:
                import lpips  # noqa: F401
            except ImportError as e:
                raise RuntimeError("eval_lpips is set but the `lpips` package (and its VGG weights) is not available") from e
            self.loss_fn_vgg = lpips.LPIPS(net="vgg").cuda()
        if cfg.test.eval_depth:  # MVSNeRF protocol, reference :23-31
            self.eval_depth_scenes = ["scan1", "scan8", "scan21", "scan103", "scan110"]
            self.depth = defaultdict(list)
        if getattr(cfg, "save_result", False):
            os.makedirs(cfg.result_dir, exist_ok=True)

    def _reset(self):
        self.psnrs, self.ssims, self.lpips = [], [], []
        self.scene = defaultdict(lambda: defaultdict(list))

    def evaluate(self, output, batch):
        B, _, _, H, W = batch["src_views"]["rgb"].shape
        gt = batch["tar_views"]["rgb"].detach().cpu().numpy()
        masks = batch["tar_views"]["mask"].cpu().numpy() >= 1
        pred = output["rgb"].permute(0, 2, 3, 1).detach().clamp(0.0, 1.0).cpu().numpy()
        if self.cfg.test.eval_center:  # LLFF protocol: drop a 10 % border   (reference :41-45)
            ch, cw = int(H * 0.1), int(W * 0.1)
            gt, pred, masks = gt[:, ch:-ch, cw:-cw], pred[:, ch:-ch, cw:-cw], masks[:, ch:-ch, cw:-cw]
        for b in range(B):
            scene = batch["meta"]["scene"][b]
            if getattr(self.cfg, "save_result", False):
                name = "{}_{}_{}.png".format(scene, batch["meta"]["tar_view"][b].item(), batch["meta"]["frame_id"][b].item())
                write_png(os.path.join(self.cfg.result_dir, name), (pred[b] * 255).clip(0, 255).astype(np.uint8))
            m = masks[b]
            g, p = gt[b].copy(), pred[b].copy()
            g[~m], p[~m] = 0.0, 0.0
            row = {"psnr": psnr(g[m], p[m], 1.0), "ssim": ssim(g, p)}
            if self.loss_fn_vgg is not None:
                import torch
                t = lambda a: (torch.from_numpy(a)[None].permute(0, 3, 1, 2) - 0.5) * 2.0
                row["lpips"] = self.loss_fn_vgg(t(g).cuda(), t(p).cuda()).item()
            for k, v in row.items():
                getattr(self, k + "s" if k != "lpips" else "lpips").append(v)
                self.scene[scene][k].append(v)
            if self.cfg.test.eval_depth and scene in self.eval_depth_scenes:
                nd, ngt = output["nerf_depth"].cpu().numpy()[b], batch["tar_views"]["depth"].cpu().numpy()[b]
                md, mgt = output["mvs_depth"].cpu().numpy()[b], batch["tar_gt_ms"]["depth"][-1][b].cpu().numpy()
                nd = _resize_bilinear(nd, ngt.shape)
                for tag, d, g_ in (("", nd, ngt), ("mvs_", md, mgt)):
                    valid = g_ != 0.0
                    err = np.abs(d[valid] - g_[valid])
                    self.depth[tag + "abs"].append(err.mean())
                    self.depth[tag + "acc_2"].append((err < 2).mean())
                    self.depth[tag + "acc_10"].append((err < 10).mean())

    def summarize(self):
        ret = {"psnr": np.mean(self.psnrs), "ssim": np.mean(self.ssims)}
        if self.loss_fn_vgg is not None:
            ret["lpips"] = np.mean(self.lpips)
        print("=" * 30)
        for scene, rows in self.scene.items():
            line = scene.ljust(16) + " psnr: {:.2f} ssim: {:.3f} ".format(np.mean(rows["psnr"]), np.mean(rows["ssim"]))
            if "lpips" in rows:
                line += "lpips:{:.3f}".format(np.mean(rows["lpips"]))
            print(line)
        print("=" * 30)
        print(ret)
        if self.cfg.test.eval_depth:
            for prefix in ("", "mvs_"):
                print({prefix + k: np.mean(self.depth[prefix + k]) for k in ("abs", "acc_2", "acc_10")})
            self.depth = defaultdict(list)
        self._reset()
        if getattr(self.cfg, "save_result", False):
            print("Save visualization results to: {}".format(self.cfg.result_dir))
        return ret
