"""Torch-facing driver of the HIP hot path.  PyTorch is plumbing here (device memory, the
current stream); all arithmetic happens in libgdbnerf_hip.so through the C ABI.

A frame is a dict of contiguous float32 CUDA tensors with the reference's layouts
(Network.forward, network.py:114-166 of the reference):
    src_images (B,V,3,Ho,Wo)   img_feat (B,V,C_f+3,H,W)   feat_volume (B,C_v,D,H,W)
    depth_range, vol_range (B,2,H,W)   src_exts (B,V,4,4)   src_ints (B,V,3,3)
    tar_ext (B,4,4)   tar_int (B,3,3)   near_far (B,2)
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional

import numpy as np
import torch

from . import _lib
from ._lib import GdbConfig, GdbFrame

NERF_KEYS = ("view_fc.0", "global_fc.0", "agg_w_fc.0", "fc.0", "lr0.0", "sigma.0", "weight.0", "weight.2", "feat_head.0")

_FRAME_SHAPES = {
    "src_images": lambda B, V, Ho, Wo, H, W, D, Cf, Cv: (B, V, 3, Ho, Wo),
    "img_feat": lambda B, V, Ho, Wo, H, W, D, Cf, Cv: (B, V, Cf + 3, H, W),
    "feat_volume": lambda B, V, Ho, Wo, H, W, D, Cf, Cv: (B, Cv, D, H, W),
    "depth_range": lambda B, V, Ho, Wo, H, W, D, Cf, Cv: (B, 2, H, W),
    "vol_range": lambda B, V, Ho, Wo, H, W, D, Cf, Cv: (B, 2, H, W),
    "src_exts": lambda B, V, Ho, Wo, H, W, D, Cf, Cv: (B, V, 4, 4),
    "src_ints": lambda B, V, Ho, Wo, H, W, D, Cf, Cv: (B, V, 3, 3),
    "tar_ext": lambda B, V, Ho, Wo, H, W, D, Cf, Cv: (B, 4, 4),
    "tar_int": lambda B, V, Ho, Wo, H, W, D, Cf, Cv: (B, 3, 3),
    "near_far": lambda B, V, Ho, Wo, H, W, D, Cf, Cv: (B, 2),
}


def _on_device(method):
    """Run an engine method with the engine's device current: the C ABI launches on a stream of that device, and HIP
    launches go to the CURRENT device (a second engine on another GPU of the same process must not inherit it)."""
    import functools

    @functools.wraps(method)
    def wrapped(self, *args, **kwargs):
        with torch.cuda.device(self.device):
            return method(self, *args, **kwargs)
    return wrapped


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _chk(t: torch.Tensor, name: str, shape=None, dtype=torch.float32) -> torch.Tensor:
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise ValueError(f"{name} must be a CUDA tensor")
    if t.dtype != dtype:
        raise ValueError(f"{name} must be {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name} must be contiguous")
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise ValueError(f"{name} has shape {tuple(t.shape)}, expected {tuple(shape)}")
    return t


class HotPathEngine:
    """One engine per (config, device).  Not thread-safe: one frame in flight per engine.

    Output buffers: by default every call returns fresh tensors.  With `reuse_outputs = True` the tensors `render`,
    `render_packed`, `decode` and `merge` return when no `out` is passed are per-engine buffers, allocated once per shape and
    OVERWRITTEN by the next call of the same method — no allocation and no memset in the per-frame path (every element is written
    by the kernels); clone what must outlive the next frame.  `reuse_internal = True` (what `Network.forward` sets) does that
    for the intermediates of a frame only — the packed render and the decoder's image, which a caller of the network never sees —
    while `merge` / `merge_packed` still return fresh tensors, as the reference's forward does (`nerf.reuse_outputs: true` turns
    the full reuse on, for loops that consume a frame's outputs before the next forward: bench.py, tools/bench_network.py)."""

    def __init__(self, *, bundle_size: int = 2, max_num_samples: int = 3, is_adaptive: bool = True,
                 inv_depth: bool = False, global_num_depth: int = 64, max_mipmap_level: int = 3,
                 feat_dim: int = 16, voxel_dim: int = 8, hid_dim: int = 64, viewdir_agg: bool = True,
                 device: str | torch.device = "cuda"):
        self.lib = _lib.load()
        self.cfg = GdbConfig(int(bundle_size), int(max_num_samples), int(bool(is_adaptive)), int(bool(inv_depth)),
                             int(global_num_depth), int(max_mipmap_level), int(feat_dim), int(voxel_dim),
                             int(hid_dim), int(bool(viewdir_agg)))
        n = C.c_size_t()
        _lib.check(self.lib.gdb_packed_weight_floats(C.byref(self.cfg), C.byref(n)))  # validates the config
        self._n_packed = n.value
        self.device = torch.device(device)
        self.weights: Optional[torch.Tensor] = None
        self._frame: Optional[GdbFrame] = None
        self._keep: Dict[str, torch.Tensor] = {}
        self._ws: Optional[torch.Tensor] = None
        self._plan_key = None
        self._pyr16_ready = False
        self._pyr32_ready = True
        self._plan_static = False
        self._plan_rows = (0, 0)
        self._fused_ok = None
        self._fpn_ptr = None
        self.f16_only_prepare = True   # a PREC_F16 engine's prepare() writes the half-precision pyramid alone (see prepare)
        # prepare(rows=strip): None = the library decides by frame size whether the strip builds only its reach of the pyramid(s)
        # (GDB_PREP_STRIP_REACH / _WHOLE: True / False force it)
        self.strip_reach: Optional[bool] = None
        self.schedule = _lib.SCHED_AUTO
        self.precision = _lib.PREC_F32  # the reference computes in fp32 (nerf.py:84-115); PREC_F16 is the opt-in fast path
        self.reuse_outputs = False
        self.reuse_internal = False
        self._bufs: Dict[tuple, torch.Tensor] = {}
        self.mip_levels: Optional[int] = None  # levels built beyond level 0 by the last prepare()
        self._warned_levels = False

    def _buf(self, name: str, shape, dtype=torch.float32, internal: bool = False) -> torch.Tensor:
        """Per-engine output buffer (see the class docstring); a fresh uninitialised tensor when reuse is off.
        internal: an intermediate of a frame (packed render, decoder image) - also reused under `reuse_internal`."""
        shape = tuple(int(x) for x in shape)
        if not (self.reuse_outputs or (internal and self.reuse_internal)):
            return torch.empty(shape, dtype=dtype, device=self.device)  # (no zero-fill either: the kernels write every element)
        key = (name, shape, dtype)
        t = self._bufs.get(key)
        if t is None:
            for k in [k for k in self._bufs if k[0] == name]:  # one shape per name: a new frame size drops the old buffer
                del self._bufs[k]
            t = self._bufs[key] = torch.empty(shape, dtype=dtype, device=self.device)
        return t

    # ---- derived sizes -------------------------------------------------------------------
    @property
    def b(self) -> int:
        return self.cfg.bundle_size

    @property
    def P(self) -> int:  # per-view channels [rgbs | feat | rgb | dir]
        return 3 * self.b * self.b + self.cfg.feat_dim + 3 + 4

    @property
    def Q(self) -> int:  # per-bundle output channels
        return 3 * self.b * self.b + self.cfg.feat_dim + 3 + self.cfg.voxel_dim

    def _stream(self) -> int:
        return torch.cuda.current_stream(self.device).cuda_stream

    # ---- weights -------------------------------------------------------------------------
    def load_weights(self, state: Dict[str, "torch.Tensor | np.ndarray"], prefix: str = "") -> None:
        """Pack the 18 NeRF tensors (keys `<prefix>view_fc.0.weight` ...) and upload them once."""
        arrs, ptrs = [], (C.c_void_p * 18)()
        for i, key in enumerate(NERF_KEYS):
            for j, suffix in enumerate((".weight", ".bias")):
                k = prefix + key + suffix
                if k not in state:
                    if key == "view_fc.0" and not self.cfg.viewdir_agg:
                        ptrs[2 * i + j] = None
                        continue
                    raise KeyError(k)
                v = state[k]
                a = np.ascontiguousarray(v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else v, dtype=np.float32)
                arrs.append(a)
                ptrs[2 * i + j] = a.ctypes.data
        expect = {"view_fc.0": (19, 4), "global_fc.0": (32, 57), "agg_w_fc.0": (1, 32), "fc.0": (16, 32), "lr0.0": (64, 24),
                  "sigma.0": (1, 64), "weight.0": (64, 111), "weight.2": (1, 64), "feat_head.0": (8, 64)}
        for key, shp in expect.items():
            k = prefix + key + ".weight"
            if k in state and tuple(state[k].shape) != shp:
                raise ValueError(f"{k} has shape {tuple(state[k].shape)}, expected {shp}")
        host = np.zeros(self._n_packed, dtype=np.float32)
        _lib.check(self.lib.gdb_pack_weights(C.byref(self.cfg), ptrs, host.ctypes.data))
        self.weights = torch.from_numpy(host).to(self.device)

    # ---- per-frame preparation -----------------------------------------------------------
    @_on_device
    def prepare(self, frame: Dict[str, torch.Tensor], im_size=None, rows=None, sources_unchanged: bool = False) -> Optional[int]:
        """Validate shapes on the host, then build the camera block and the feature pyramid.  A frame
        without the source side (only tar_ext, tar_int, near_far [, depth_range, vol_range]; pass
        `im_size=(Ho,Wo)`) prepares the target camera alone — enough for build_rays / sample.
        Returns the number of mip levels built beyond level 0 (None without the source side); it is below
        `max_mipmap_level` when an extent of the feature map turns odd on the way down (warned once).
        rows = (r0, r1): this engine will render the bundle-map rows [r0, r1) only (a rank's strip, `parallel.row_strip`): the list
        schedules' plan is built for those rows alone (gdb_prepare_rows); a render outside them rebuilds the plan itself.
        sources_unchanged = True: the caller vouches that `src_images` and `img_feat` / `fpn_feat` are the tensors - same storage, same
        contents - of this engine's previous prepare() (a sweep of target views over fixed source views): the feature pyramid(s) and the
        half-precision image copy in the workspace are kept (GDB_PREP_SOURCES_READY, ABI v7), the camera block and the plan are rebuilt.
        Ignored (a full prepare runs) when the workspace holds no such products: first call, another frame shape, another precision's
        products, or other storage pointers than last time."""
        b = self.b
        if "src_images" in frame:
            si = frame["src_images"]
            if si.dim() != 5:
                raise ValueError("src_images must be (B,V,3,Ho,Wo)")
            B, V, _, Ho, Wo = si.shape
            D = frame["feat_volume"].shape[2]
        else:
            (Ho, Wo), B, V, D = im_size, frame["tar_ext"].shape[0], 1, 1
        if Ho % b or Wo % b:
            raise ValueError(f"image {Ho}x{Wo} not divisible by bundle_size {b}")
        H, W = Ho // b, Wo // b
        dims = (B, V, Ho, Wo, H, W, D, self.cfg.feat_dim, self.cfg.voxel_dim)
        fpn = frame.get("fpn_feat")  # N3: the FPN level alone, (B,V,C_f,H,W); the colour channels are resampled by the kernel
        if fpn is not None:
            if "img_feat" in frame:
                raise ValueError("give img_feat (features + colours) or fpn_feat (features alone), not both")
            _chk(fpn, "fpn_feat", (B, V, self.cfg.feat_dim, H, W))
        for name, shp in _FRAME_SHAPES.items():
            if name in frame:
                _chk(frame[name], name, shp(*dims))
            elif name == "img_feat" and fpn is not None:
                continue
            elif "src_images" in frame or name in ("tar_ext", "tar_int", "near_far"):
                raise KeyError(name)
        f = GdbFrame(B, V, Ho, Wo, H, W, D, *(_ptr(frame.get(k)) for k in (
            "src_images", "img_feat", "feat_volume", "depth_range", "vol_range", "src_exts", "src_ints",
            "tar_ext", "tar_int", "near_far")))
        need = C.c_size_t()
        _lib.check(self.lib.gdb_workspace_bytes(C.byref(self.cfg), C.byref(f), C.byref(need)))
        if self._ws is None or self._ws.numel() < need.value:
            self._ws = torch.empty(need.value, dtype=torch.uint8, device=self.device)
        self._frame, self._keep = f, dict(frame)
        self._fused_ok = None
        # gdb_prepare builds the dense schedule's plan from the depth prior as it is NOW (adaptive configs); a later render may
        # skip its own rebuild (GDB_SCHED_PLAN_READY) only while that tensor is unchanged: same storage, same version counter
        self._plan_key = None   # armed only once gdb_prepare has returned OK (below)
        if "src_images" in frame:
            lay = (C.c_size_t * 7)()
            _lib.check(self.lib.gdb_pyramid_layout(C.byref(self.cfg), C.byref(f), lay))
            self.mip_levels = int(lay[2])
            if self.mip_levels < min(self.cfg.max_mipmap_level, 3) and not self._warned_levels:
                import warnings
                self._warned_levels = True
                # DESIGN.md §2, "odd-extent rule": a level is built only while both extents of the one below are even and >= 2
                warnings.warn(f"feature map {H}x{W}: the mip chain stops at level {self.mip_levels} (an extent turns odd), below "
                              f"nerf.max_mipmap_level = {self.cfg.max_mipmap_level}; footprints are clamped to that level")
        # an engine whose precision is PREC_F16 has the half-precision copy of the pyramid written in the same launch (the f16 render
        # gathers from it); any other engine that is asked for an f16 render later lets that render convert the fp32 pyramid itself
        # ... and ONLY that copy (`f16_only_prepare`, default on): the fp32 pyramid is 26 of k_prepare's 58 MB of traffic at the DTU
        # size and an f16 render never reads it.  Whatever does read it later on this frame - an f32 / split-f16 render, the
        # operator mirror encode(), feature_pyramid() - goes through _need_pyr32(), which prepares again in full first.
        self._pyr16_ready = False
        self._pyr32_ready = False
        self._fpn_ptr = None if fpn is None else fpn.data_ptr()
        flags = _lib.PREP_PYR16 if (self.precision == _lib.PREC_F16 and "src_images" in frame) else 0
        if flags and self.f16_only_prepare:
            flags |= _lib.PREP_PYR16_ONLY
        # what the source-only products in the workspace were built from: shape, flags, storage pointers (contents are the caller's promise)
        src_key = None
        if "src_images" in frame:
            src_key = (dims, flags, frame["src_images"].data_ptr(), (fpn if fpn is not None else frame["img_feat"]).data_ptr(), self._ws.data_ptr())
        if sources_unchanged and src_key is not None and src_key == getattr(self, "_src_key", None):
            flags |= _lib.PREP_SOURCES_READY
        self._src_key = None   # armed again once the launch below has been accepted
        # a row strip smaller than the frame builds only the part of the pyramid(s) its samples can reach (gdb_prepare_rows, round 6):
        # renders of other rows, the operator mirrors and feature_pyramid() then prepare again in full first (_need_whole)
        partial = rows is not None and (int(rows[0]) > 0 or int(rows[1]) < H) and "src_images" in frame and not (flags & _lib.PREP_SOURCES_READY)
        self._plan_rows = (0, H) if rows is None else (int(rows[0]), int(rows[1]))
        if rows is None:
            _lib.check(self.lib.gdb_prepare_ex(C.byref(self.cfg), C.byref(f), self._fpn_ptr, flags,
                                               self._ws.data_ptr(), self._ws.numel(), self._stream()))
        else:
            sflag = 0 if self.strip_reach is None else (_lib.PREP_STRIP_REACH if self.strip_reach else _lib.PREP_STRIP_WHOLE)
            partial = partial and self.strip_reach is not False   # (by size: the engine cannot know - it assumes the strip's reach alone)
            _lib.check(self.lib.gdb_prepare_rows(C.byref(self.cfg), C.byref(f), self._fpn_ptr, flags | sflag, self._plan_rows[0], self._plan_rows[1],
                                                 self._ws.data_ptr(), self._ws.numel(), self._stream()))
        self._src_key = None if partial else src_key   # (a partial pyramid is no basis for a later sources_unchanged prepare)
        self._pyr_partial = partial
        self._pyr16_ready = bool(flags & _lib.PREP_PYR16)
        self._pyr32_ready = not (flags & _lib.PREP_PYR16_ONLY)
        if self.cfg.is_adaptive:
            self._plan_key = self._prior_key(frame.get("depth_range"))
        # fixed counts: whether gdb_prepare built the list schedules' plan is the LIBRARY's rule (gdb_render_info out[2]), asked, not restated
        self._plan_static = (not self.cfg.is_adaptive and "depth_range" in frame and "src_images" in frame
                             and bool(self.render_info()["plan_built_by_prepare"]))
        return self.mip_levels

    def render_info(self, precision: Optional[int] = None, row_begin: int = 0, row_end: Optional[int] = None) -> Dict[str, int]:
        """What a fused render of the frame last prepared would do, as the library itself answers (gdb_render_info, ABI v6):
        `fused` (0: only the operator mirrors run this config - bundle_size 1 / 4, a single view), `schedule` (what GDB_SCHED_AUTO
        resolves to: 1 slot waves, 2 segment wave, 3 dense, 4 flat), `plan_built_by_prepare`, `launches`, `kernel` (its name in a trace)."""
        f = self._need_frame()
        out = (C.c_int32 * 4)()
        precision = self.precision if precision is None else precision
        _lib.check(self.lib.gdb_render_info(C.byref(self.cfg), C.byref(f), int(precision), int(row_begin), int(f.H if row_end is None else row_end), out))
        sched = int(out[1]) if self.cfg.bundle_size != 2 else (int(self.schedule) or int(out[1]))   # (bundle_size 1 / 4: the one schedule, whatever the engine's setting)
        return {"fused": int(out[0]), "schedule": int(out[1]), "plan_built_by_prepare": int(out[2]), "launches": int(out[3]),
                "kernel": {0: None, 1: "k_render_fused", 2: "k_render_solo", 3: "k_render_dense", 4: "k_render_flat"}[sched if out[0] else 0]}

    @property
    def fused_supported(self) -> bool:
        """Whether gdb_render_bundles_fused / _packed accept this engine's config and the frame last prepared (asked once per prepare)."""
        if self._fused_ok is None:
            self._fused_ok = bool(self.render_info()["fused"])
        return self._fused_ok

    def _need_whole(self, row_begin: int = 0, row_end: Optional[int] = None) -> None:
        """The pyramid(s) of the WHOLE frame: after a prepare(rows=strip) that built only the strip's reach (`_pyr_partial`), anything that
        looks beyond the strip - a render of other rows, the operator mirrors, feature_pyramid() - prepares the kept frame again in full."""
        if not getattr(self, "_pyr_partial", False) or self._frame is None:
            return
        p0, p1 = self._plan_rows
        if row_begin >= p0 and (self._frame.H if row_end is None else row_end) <= p1 and row_end is not None:
            return
        self.prepare(self._keep)

    def _need_pyr32(self) -> None:
        """The fp32 feature pyramid of the frame last prepared, built now if that prepare wrote the half-precision copy alone."""
        self._need_whole()
        if getattr(self, "_pyr32_ready", True) or self._frame is None:
            return
        flags = _lib.PREP_PYR16 if self._pyr16_ready else 0
        _lib.check(self.lib.gdb_prepare_rows(C.byref(self.cfg), C.byref(self._frame), self._fpn_ptr, flags, self._plan_rows[0], self._plan_rows[1],
                                             self._ws.data_ptr(), self._ws.numel(), self._stream()))
        self._pyr32_ready = True
        self._src_key = None   # (the workspace now holds other products than the key describes: the next prepare() runs in full)

    def _need_pyr32_strip(self, row_begin: int, row_end: int) -> None:
        """_need_pyr32 for a render of rows that lie inside a partial prepare's strip: the fp32 pyramid of that strip's reach suffices."""
        if getattr(self, "_pyr32_ready", True) or self._frame is None:
            return
        if getattr(self, "_pyr_partial", False):   # (rows inside the strip: _need_whole has returned without preparing)
            flags = _lib.PREP_PYR16 if self._pyr16_ready else 0
            flags |= 0 if self.strip_reach is None else _lib.PREP_STRIP_REACH
            _lib.check(self.lib.gdb_prepare_rows(C.byref(self.cfg), C.byref(self._frame), self._fpn_ptr, flags, self._plan_rows[0], self._plan_rows[1],
                                                 self._ws.data_ptr(), self._ws.numel(), self._stream()))
            self._pyr32_ready = True
            return
        self._need_pyr32()

    @staticmethod
    def _prior_key(dr: Optional[torch.Tensor]):
        """(storage pointer, version counter) of the depth prior, or None when the tensor has no version counter to watch (tensors
        created under torch.inference_mode(): `_version` raises) - the render then rebuilds the plan itself on every call (k_plan,
        a ~1 us launch).  Writes that bypass the counter (`.data.copy_`, an external kernel writing through data_ptr) are not seen:
        call `invalidate_plan()` after one."""
        if dr is None:
            return None
        try:
            if dr.is_inference():
                return None
            return (dr.data_ptr(), dr._version)
        except (RuntimeError, AttributeError):   # (`_version` is a private attribute: a torch without it degrades to "rebuild the plan")
            return None

    def invalidate_plan(self) -> None:
        """Forget that the last prepare() built the dense schedule's plan: the next render rebuilds it from the depth prior as it
        stands (use after changing `depth_range` in a way torch's version counter does not see)."""
        self._plan_key = None

    # ---- next row N1: merge around the decoder ------------------------------------------------
    @_on_device
    def merge(self, bundle_feat, rgb_c=None, bundle_depth=None, bundle_opacity=None, reweighting: bool = False):
        """network.py:170-182 on the frame last prepared: img = rgb_c + pixel_shuffle(bundle_feat[:, :3 b^2], b) (re-weighted
        if asked), bundle depth / opacity maps upsampled x b.  Returns (img (B,3,Ho,Wo), depth (B,Ho,Wo) | None, opacity | None)."""
        f = self._need_frame()
        nb, b = self.n_bundles, self.cfg.bundle_size
        Ho, Wo = f.H * b, f.W * b
        _chk(bundle_feat, "bundle_feat", (nb, self.Q))
        if rgb_c is not None:
            _chk(rgb_c, "rgb_c", (f.B, 3, Ho, Wo))
        img = self._buf("merge.img", (f.B, 3, Ho, Wo))
        outs = []
        for name, m in (("bundle_depth", bundle_depth), ("bundle_opacity", bundle_opacity)):
            if m is not None:
                _chk(m, name, (nb,))
                outs.append(self._buf("merge." + name, (f.B, Ho, Wo)))
            else:
                outs.append(None)
        p = lambda t: None if t is None else t.data_ptr()
        _lib.check(self.lib.gdb_merge(C.byref(self.cfg), C.byref(f), bundle_feat.data_ptr(), p(rgb_c), p(bundle_depth), p(bundle_opacity),
                                      int(bool(reweighting)), img.data_ptr(), p(outs[0]), p(outs[1]), self._stream()))
        return img, outs[0], outs[1]

    @_on_device
    def merge_packed(self, packed: torch.Tensor, rgb_c: Optional[torch.Tensor] = None, reweighting: bool = False):
        """`merge` on the packed render ((n_bundles, Q + 2) rows [bundle_feat | depth | opacity], read in place): the buffer
        `render_packed` fills and a row-strip all-gather (parallel.StripGather) leaves on every rank.
        Returns (img (B,3,Ho,Wo), depth (B,Ho,Wo), opacity (B,Ho,Wo))."""
        f = self._need_frame()
        nb, b = self.n_bundles, self.cfg.bundle_size
        Ho, Wo = f.H * b, f.W * b
        _chk(packed, "packed", (nb, self.Q + 2))
        if rgb_c is not None:
            _chk(rgb_c, "rgb_c", (f.B, 3, Ho, Wo))
        img, dep, opa = self._buf("merge.img", (f.B, 3, Ho, Wo)), self._buf("merge.bundle_depth", (f.B, Ho, Wo)), self._buf("merge.bundle_opacity", (f.B, Ho, Wo))
        _lib.check(self.lib.gdb_merge_packed(C.byref(self.cfg), C.byref(f), packed.data_ptr(), _ptr(rgb_c), int(bool(reweighting)),
                                             img.data_ptr(), dep.data_ptr(), opa.data_ptr(), self._stream()))
        return img, dep, opa

    # ---- next row N1: the decoder -----------------------------------------------------------------
    def load_decoder_weights(self, state: Dict[str, "torch.Tensor | np.ndarray"], num_layers: int, prefix: str = "") -> None:
        """Pack the Decoder's tensors (keys `<prefix>in_conv.weight` ... as in decoder_rdn.py:55-65) and upload them once."""
        keys = ["in_conv.weight", "in_conv.bias"]
        for i in range(num_layers):
            keys += [f"blocks.{i}.conv1.weight", f"blocks.{i}.conv2.weight", f"blocks.{i}.conv3.weight", f"blocks.{i}.se.fc.0.weight", f"blocks.{i}.se.fc.2.weight"]
        keys += ["up.0.weight", "up.0.bias"]
        if self.cfg.bundle_size == 4:   # upscale_factor 4: a second up stage (decoder_rdn.py:59-62: nn.Sequential indices 0 and 2 are the convolutions)
            keys += ["up.2.weight", "up.2.bias"]
        keys += ["out_conv.weight", "out_conv.bias"]
        cin = self.cfg.feat_dim + 3 + self.cfg.voxel_dim
        expect = {"in_conv.weight": (64, cin, 3, 3), "up.0.weight": (256, 64, 3, 3), "up.2.weight": (256, 64, 3, 3), "out_conv.weight": (3, 64, 1, 1)}
        arrs, ptrs = [], (C.c_void_p * len(keys))()
        for i, k in enumerate(keys):
            v = state[prefix + k]
            a = np.ascontiguousarray(v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else v, dtype=np.float32)
            if k in expect and tuple(a.shape) != expect[k]:
                raise ValueError(f"{prefix + k} has shape {tuple(a.shape)}, expected {expect[k]}")
            arrs.append(a)
            ptrs[i] = a.ctypes.data
        n = C.c_size_t()
        _lib.check(self.lib.gdb_decoder_packed_floats(C.byref(self.cfg), int(num_layers), C.byref(n)))
        host = np.zeros(n.value, dtype=np.float32)
        _lib.check(self.lib.gdb_pack_decoder_weights(C.byref(self.cfg), int(num_layers), ptrs, host.ctypes.data))
        self.dec_weights, self.dec_layers = torch.from_numpy(host).to(self.device), int(num_layers)

    @_on_device
    def decode(self, bundle_feat: torch.Tensor, precision: Optional[int] = None) -> torch.Tensor:
        """Decoder.forward (decoder_rdn.py:67-81) on the frame last prepared: bundle_feat (n_bundles, Q) or the packed
        (n_bundles, Q + 2) render, read in place (channels 3b^2 .. Q-1 are the decoder's input, network.py:170-174) ->
        rgb_c (B, 3, Ho, Wo).  precision: None = fp32 MFMA unless this engine's `precision` is PREC_F32X or PREC_F16 (then the
        split-f16 convolutions, fp32-grade), 1 = fp32 MFMA, 2 = split-f16."""
        if precision is None:
            precision = _lib.PREC_F32 if self.precision == _lib.PREC_F32 else _lib.PREC_F32X
        if getattr(self, "dec_weights", None) is None:
            raise ValueError("load_decoder_weights() first")
        f = self._need_frame()
        nb = self.n_bundles
        if bundle_feat.dim() != 2 or bundle_feat.shape[0] != nb or bundle_feat.shape[1] < self.Q:
            raise ValueError(f"bundle_feat has shape {tuple(bundle_feat.shape)}, expected ({nb}, >= {self.Q})")
        _chk(bundle_feat, "bundle_feat")
        need = C.c_size_t()
        _lib.check(self.lib.gdb_decoder_workspace_bytes(C.byref(self.cfg), C.byref(f), C.byref(need)))
        if getattr(self, "_dec_ws", None) is None or self._dec_ws.numel() < need.value:
            self._dec_ws = torch.empty(need.value, dtype=torch.uint8, device=self.device)
        b = self.cfg.bundle_size
        rgb_c = self._buf("decode.rgb_c", (f.B, 3, f.H * b, f.W * b), internal=True)
        _lib.check(self.lib.gdb_decode(C.byref(self.cfg), C.byref(f), bundle_feat.data_ptr(), int(bundle_feat.shape[1]), self.dec_weights.data_ptr(),
                                       self.dec_layers, int(precision), self._dec_ws.data_ptr(), self._dec_ws.numel(), rgb_c.data_ptr(), self._stream()))
        return rgb_c

    def set_schedule(self, mode: int) -> None:
        """Work decomposition of THIS engine's fused calls (a per-call argument of the C ABI, no process-wide state):
        0 auto, 1 one wave per sample slot, 2 one wave per 32-bundle segment, 3 dense (one wave per window of whole bundles holding
        <= 32 samples of the compacted sample list), 4 flat (one wave per 32 consecutive samples of it; bundles may straddle windows).
        See include/gdb_nerf_hip.h."""
        if int(mode) not in (0, 1, 2, 3, 4):
            raise ValueError(f"schedule {mode} outside 0..4")
        self.schedule = int(mode)

    def feature_pyramid(self):
        """The mip pyramid `prepare` built, as nvdiffrast would hold it: a list over levels of
        (B, V, H_l, W_l, C_f+3) tensors (copies; the workspace keeps the chunk-planar layout)."""
        f = self._need_frame()
        self._need_pyr32()
        out = (C.c_size_t * 7)()
        _lib.check(self.lib.gdb_pyramid_layout(C.byref(self.cfg), C.byref(f), out))
        off, stride, levels = out[0], out[1], out[2]
        n = f.B * f.V
        pyr = self._ws[off:off + 4 * stride * n].view(torch.float32).view(n, stride)
        res = []
        for l in range(levels + 1):
            h, w = f.H >> l, f.W >> l
            lv = pyr[:, out[3 + l]:out[3 + l] + 20 * h * w].reshape(n, 5, h, w, 4)       # [chunk][y][x][4]
            res.append(lv.permute(0, 2, 3, 1, 4).reshape(f.B, f.V, h, w, 20)[..., :19].contiguous())
        return res

    def feature_pyramid16(self):
        """The HALF-precision copy of the pyramid (what a PREC_F16 render gathers from), per level (B, V, H_l, W_l, C_f+3) float16
        (copies).  Only meaningful after a prepare() of a PREC_F16 engine or after an f16 render (which builds it on demand)."""
        f = self._need_frame()
        self._need_whole()
        out = (C.c_size_t * 7)()
        _lib.check(self.lib.gdb_pyramid16_layout(C.byref(self.cfg), C.byref(f), out))
        off, stride, levels = out[0], out[1], out[2]
        n = f.B * f.V
        blk = self._ws[off:off + stride * n].view(n, stride)
        res = []
        for l in range(levels + 1):
            h, w = f.H >> l, f.W >> l
            hw, lo = h * w, out[3 + l]
            p0 = blk[:, lo:lo + 16 * hw].contiguous().view(torch.float16).view(n, hw, 8)            # channels 0..3, 8..11
            p1 = blk[:, lo + 16 * hw:lo + 32 * hw].contiguous().view(torch.float16).view(n, hw, 8)  # channels 4..7, 12..15
            p2 = blk[:, lo + 32 * hw:lo + 40 * hw].contiguous().view(torch.float16).view(n, hw, 4)  # channels 16..19
            lv = torch.cat((p0[..., :4], p1[..., :4], p0[..., 4:], p1[..., 4:], p2), dim=-1)        # (n, hw, 20)
            res.append(lv.view(f.B, f.V, h, w, 20)[..., :19].contiguous())
        return res

    def dense_plan(self) -> torch.Tensor:
        """The dense schedule's plan built from the depth prior: (B*H, stride) int32, row = [n_windows, first sample offset of each
        window, the row's sample total]; plus L = 33 - S_max (the fixed-cut window length) as `.window` attribute.  See
        include/gdb_nerf_hip.h gdb_dense_plan_layout."""
        f = self._need_frame()
        out = (C.c_size_t * 3)()
        _lib.check(self.lib.gdb_dense_plan_layout(C.byref(self.cfg), C.byref(f), out))
        rows = f.B * f.H
        t = self._ws[out[0]:out[0] + 4 * rows * out[1]].view(torch.int32).view(rows, out[1]).clone()
        t.window = int(out[2])
        return t

    def dense_map(self) -> torch.Tensor:
        """The compacted sample list beside the plan: (B*H, stride) int64 (from uint32), entry s of a row = bundle | slot << 16 |
        count << 24 of the sample at offset s, 0xFFFFFFFF past the row's last sample (include/gdb_nerf_hip.h gdb_dense_map_layout)."""
        f = self._need_frame()
        out = (C.c_size_t * 2)()
        _lib.check(self.lib.gdb_dense_map_layout(C.byref(self.cfg), C.byref(f), out))
        rows = f.B * f.H
        t = self._ws[out[0]:out[0] + 4 * rows * out[1]].view(torch.int32).view(rows, out[1]).to(torch.int64)
        return t & 0xFFFFFFFF

    def _need_frame(self) -> GdbFrame:
        if self._frame is None:
            # bundle_sampler.py:220-221 of the reference
            raise ValueError("Rays have not been built yet. Please call prepare() first.")
        return self._frame

    def _sched(self, row_begin: int = 0, row_end: Optional[int] = None) -> int:
        """The schedule argument of a render call: this engine's schedule, plus the plan-is-current flag while the depth prior
        the last prepare() consumed is untouched and the strip lies inside the rows that prepare planned."""
        ready = self._plan_key is not None and self._prior_key(self._keep.get("depth_range")) == self._plan_key
        if not self.cfg.is_adaptive:   # fixed counts: the plan gdb_prepare built (gdb_fixed_counts_dense) does not depend on the prior's values
            ready = self._plan_static
        p0, p1 = self._plan_rows
        ready = ready and row_begin >= p0 and (self._frame.H if row_end is None else row_end) <= p1
        return int(self.schedule) | (_lib.SCHED_PLAN_READY if ready else 0) | (_lib.SCHED_PYR16_READY if self._pyr16_ready else 0)

    @property
    def n_bundles(self) -> int:
        f = self._need_frame()
        return f.B * f.H * f.W

    # ---- operator mirrors ----------------------------------------------------------------
    @_on_device
    def build_rays(self) -> Dict[str, torch.Tensor]:
        f = self._need_frame()
        dev = self.device
        out = {"rays_d": torch.empty((f.B, f.Ho, f.Wo, 3), device=dev), "uv": torch.empty((f.Ho, f.Wo, 2), device=dev),
               "rays_o": torch.empty((f.B, 3), device=dev), "z_axis": torch.empty((f.B, 3), device=dev),
               "tar_pixel_radius": torch.empty((f.B,), device=dev)}
        _lib.check(self.lib.gdb_build_rays(C.byref(self.cfg), C.byref(f), self._ws.data_ptr(), *(out[k].data_ptr() for k in (
            "rays_d", "uv", "rays_o", "z_axis", "tar_pixel_radius")), self._stream()))
        return out

    @_on_device
    def sample(self) -> Dict[str, torch.Tensor]:
        """Arrays are allocated for N_max = n_bundles * S_max; `total` (device int64) holds the valid count."""
        f = self._need_frame()
        dev, bb = self.device, self.b * self.b
        nmax = self.n_bundles * self.cfg.max_num_samples
        out = {"rays_xyz": torch.empty((nmax, 3, bb), device=dev), "uvd": torch.empty((nmax, 3), device=dev),
               "z_vals": torch.empty((nmax,), device=dev), "ball_radii": torch.empty((nmax,), device=dev),
               "indices": torch.empty((nmax,), dtype=torch.int64, device=dev),
               "samples_per_batch": torch.empty((f.B,), dtype=torch.int64, device=dev),
               "samples_per_bundle": torch.empty((self.n_bundles,), dtype=torch.int32, device=dev),
               "total": torch.empty((1,), dtype=torch.int64, device=dev)}
        _lib.check(self.lib.gdb_sample(C.byref(self.cfg), C.byref(f), self._ws.data_ptr(), *(out[k].data_ptr() for k in (
            "rays_xyz", "uvd", "z_vals", "ball_radii", "indices", "samples_per_batch", "samples_per_bundle", "total")),
            self._stream()))
        return out

    @_on_device
    def encode(self, rays_xyz: torch.Tensor, uvd: torch.Tensor, ball_radii: torch.Tensor,
               samples_per_batch: torch.Tensor, total: torch.Tensor):
        f = self._need_frame()
        self._need_pyr32()
        n = rays_xyz.shape[0]
        bb = self.b * self.b
        _chk(rays_xyz, "rays_xyz", (n, 3, bb)); _chk(uvd, "uvd", (n, 3)); _chk(ball_radii, "ball_radii", (n,))
        _chk(samples_per_batch, "samples_per_batch", (f.B,), torch.int64); _chk(total, "total", (1,), torch.int64)
        rfd = torch.empty((f.V, n, self.P), device=self.device)
        vox = torch.empty((n, self.cfg.voxel_dim), device=self.device)
        _lib.check(self.lib.gdb_encode(C.byref(self.cfg), C.byref(f), self._ws.data_ptr(), rays_xyz.data_ptr(), uvd.data_ptr(),
                                       ball_radii.data_ptr(), samples_per_batch.data_ptr(), total.data_ptr(), n,
                                       rfd.data_ptr(), vox.data_ptr(), self._stream()))
        return rfd, vox

    @_on_device
    def mlp(self, vox_feat: torch.Tensor, rgbs_feat_dir: torch.Tensor, total: Optional[torch.Tensor] = None):
        if self.weights is None:
            raise ValueError("load_weights() first")
        V, n, P = rgbs_feat_dir.shape
        _chk(rgbs_feat_dir, "rgbs_feat_dir", (V, n, self.P)); _chk(vox_feat, "vox_feat", (n, self.cfg.voxel_dim))
        if total is not None:
            _chk(total, "total", (1,), torch.int64)
        sigma = torch.empty((n,), device=self.device)
        feat = torch.empty((n, self.Q), device=self.device)
        _lib.check(self.lib.gdb_mlp(C.byref(self.cfg), self.weights.data_ptr(), V, vox_feat.data_ptr(), rgbs_feat_dir.data_ptr(),
                                    _ptr(total), n, sigma.data_ptr(), feat.data_ptr(), self._stream()))
        return sigma, feat

    @_on_device
    def composite(self, sigma: torch.Tensor, feat: torch.Tensor, z_vals: torch.Tensor, indices: torch.Tensor,
                  n_bundles: int, total: Optional[torch.Tensor] = None):
        n, ch = feat.shape
        _chk(sigma, "sigma", (n,)); _chk(feat, "feat", (n, ch)); _chk(z_vals, "z_vals", (n,))
        _chk(indices, "indices", (n,), torch.int64)
        if total is not None:
            _chk(total, "total", (1,), torch.int64)
        dev = self.device
        weights = torch.zeros((n,), device=dev)
        bf = torch.empty((n_bundles, ch), device=dev)
        depth = torch.empty((n_bundles,), device=dev)
        opac = torch.empty((n_bundles,), device=dev)
        scratch = torch.empty((2 * n_bundles,), dtype=torch.int32, device=dev)
        _lib.check(self.lib.gdb_composite(C.byref(self.cfg), sigma.data_ptr(), feat.data_ptr(), z_vals.data_ptr(),
                                          indices.data_ptr(), _ptr(total), n, n_bundles, ch, weights.data_ptr(), bf.data_ptr(),
                                          depth.data_ptr(), opac.data_ptr(), scratch.data_ptr(), self._stream()))
        return weights, bf, depth, opac

    @_on_device
    def render_weights(self, sigma: torch.Tensor, indices: torch.Tensor, n_bundles: int, total: Optional[torch.Tensor] = None):
        n = sigma.shape[0]
        _chk(sigma, "sigma", (n,)); _chk(indices, "indices", (n,), torch.int64)
        weights = torch.zeros((n,), device=self.device)
        scratch = torch.empty((2 * n_bundles,), dtype=torch.int32, device=self.device)
        _lib.check(self.lib.gdb_render_weights(C.byref(self.cfg), sigma.data_ptr(), indices.data_ptr(), _ptr(total), n, n_bundles,
                                               weights.data_ptr(), scratch.data_ptr(), self._stream()))
        return weights

    @_on_device
    def accumulate(self, weights: torch.Tensor, feat: torch.Tensor, z_vals: torch.Tensor, indices: torch.Tensor, n_bundles: int,
                   total: Optional[torch.Tensor] = None):
        n, ch = feat.shape
        _chk(weights, "weights", (n,)); _chk(feat, "feat", (n, ch)); _chk(z_vals, "z_vals", (n,)); _chk(indices, "indices", (n,), torch.int64)
        dev = self.device
        fm, dm, om = torch.empty((n_bundles, ch), device=dev), torch.empty((n_bundles,), device=dev), torch.empty((n_bundles,), device=dev)
        scratch = torch.empty((2 * n_bundles,), dtype=torch.int32, device=dev)
        _lib.check(self.lib.gdb_accumulate(C.byref(self.cfg), weights.data_ptr(), feat.data_ptr(), z_vals.data_ptr(), indices.data_ptr(),
                                           _ptr(total), n, n_bundles, ch, fm.data_ptr(), dm.data_ptr(), om.data_ptr(), scratch.data_ptr(),
                                           self._stream()))
        return fm, dm, om

    def render_unfused(self):
        """build_rays → sample → encode → MLP → composite through the operator mirrors (all fp32)."""
        s = self.sample()
        rfd, vox = self.encode(s["rays_xyz"], s["uvd"], s["ball_radii"], s["samples_per_batch"], s["total"])
        sigma, feat = self.mlp(vox, rfd, s["total"])
        _, bf, depth, opac = self.composite(sigma, feat, s["z_vals"], s["indices"], self.n_bundles, s["total"])
        return bf, depth, opac

    def render_unfused_packed(self, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """The operator-mirror chain into the packed (n_bundles, Q + 2) rows [bundle_feat | depth | opacity] `render_packed` returns:
        what `Network.forward` takes when the fused entries refuse the config (bundle_size 1 / 4; `fused_supported`)."""
        bf, depth, opac = self.render_unfused()
        nb = self.n_bundles
        if out is None:
            out = self._buf("render.packed", (nb, self.Q + 2), internal=True)
        _chk(out, "out", (nb, self.Q + 2))
        out[:, :self.Q] = bf
        out[:, self.Q] = depth
        out[:, self.Q + 1] = opac
        return out

    # ---- production entry ----------------------------------------------------------------
    @_on_device
    def render(self, row_begin: int = 0, row_end: Optional[int] = None, precision: Optional[int] = None, out=None):
        """Fused hot path over bundle-map rows [row_begin,row_end) of every batch item.  precision: None = this engine's
        default (`self.precision`), 0 = f16 MFMA operands, 1 = fp32 MFMA (the reference's precision), 2 = split-f16 operands
        (hi + lo pairs, three f16 MFMAs per product: fp32-grade, not bit-exact fp32)."""
        if self.weights is None:
            raise ValueError("load_weights() first")
        f = self._need_frame()
        row_end = f.H if row_end is None else row_end
        precision = self.precision if precision is None else precision
        self._need_whole(row_begin, row_end)
        if precision != _lib.PREC_F16:
            self._need_pyr32_strip(row_begin, row_end)
        nb = self.n_bundles
        if out is None:
            if row_begin == 0 and row_end == f.H:  # every row is written: a reused buffer needs no zero-fill
                out = (self._buf("render.bf", (nb, self.Q)), self._buf("render.depth", (nb,)), self._buf("render.opac", (nb,)))
            else:  # a strip into fresh tensors: rows outside it read as zeros
                out = (torch.zeros((nb, self.Q), device=self.device), torch.zeros((nb,), device=self.device),
                       torch.zeros((nb,), device=self.device))
        bf, depth, opac = out
        _chk(bf, "bundle_feat", (nb, self.Q)); _chk(depth, "depth", (nb,)); _chk(opac, "opacity", (nb,))
        _lib.check(self.lib.gdb_render_bundles_fused(C.byref(self.cfg), C.byref(f), self._ws.data_ptr(), self.weights.data_ptr(),
                                                     int(row_begin), int(row_end), int(precision), self._sched(row_begin, row_end), bf.data_ptr(),
                                                     depth.data_ptr(), opac.data_ptr(), self._stream()))
        return bf, depth, opac

    @_on_device
    def render_packed(self, row_begin: int = 0, row_end: Optional[int] = None, precision: Optional[int] = None,
                      out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """The same into ONE (n_bundles, Q + 2) tensor, row = [bundle_feat | depth | opacity]: a row strip is one contiguous
        block, which makes the multi-GPU exchange a single all-gather (parallel.StripGather: render the rank's strip straight
        into `StripGather.full`, then `gather()`)."""
        if self.weights is None:
            raise ValueError("load_weights() first")
        f = self._need_frame()
        row_end = f.H if row_end is None else row_end
        precision = self.precision if precision is None else precision
        self._need_whole(row_begin, row_end)
        if precision != _lib.PREC_F16:
            self._need_pyr32_strip(row_begin, row_end)
        nb = self.n_bundles
        if out is None:
            full = row_begin == 0 and row_end == f.H
            out = self._buf("render.packed", (nb, self.Q + 2), internal=True) if full else torch.zeros((nb, self.Q + 2), device=self.device)
        _chk(out, "out", (nb, self.Q + 2))
        _lib.check(self.lib.gdb_render_bundles_packed(C.byref(self.cfg), C.byref(f), self._ws.data_ptr(), self.weights.data_ptr(),
                                                      int(row_begin), int(row_end), int(precision), self._sched(row_begin, row_end),
                                                      out.data_ptr(), self._stream()))
        return out
