"""Whole-network fixture F7: the reference's own `Network` (random init, eval mode, CPU) on one 64x96
frame with 3 source views.  Runs only in the authoring container.  As in make_golden.py, the absent
third-party modules are placeholders that route their three calls to the oracle's restatements
(so `rgb` depends on them: parity unpinned for those ops, pinned for everything else).

Weights are rounded to float16-representable values before the run so that the committed state dict
is half the size; both sides then use exactly those values in float32."""
import os
import sys
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (placeholders + oracle wiring)

sys.path.insert(0, "/root/reference")
warnings.filterwarnings("ignore")

from gdb_nerf_amd import synthetic  # noqa: E402
from gdb_nerf_amd.configs import make_cfg  # noqa: E402


def main():
    mg._placeholders()
    from networks.gdb_nerf.network import Network as RefNetwork  # the reference's own class
    cfg = make_cfg("configs/dtu_eval.yaml")
    torch.manual_seed(0)
    net = RefNetwork(cfg).eval()
    with torch.no_grad():
        for p in net.parameters():
            p.copy_(p.half().float())
        for m in net.modules():  # give the batch norms non-trivial running statistics
            if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
                m.running_mean.uniform_(-0.1, 0.1)
                m.running_var.uniform_(0.8, 1.2)
                m.running_mean.copy_(m.running_mean.half().float())
                m.running_var.copy_(m.running_var.half().float())
    frame = synthetic.make_frame(64, 96, V=3, B=1, seed=9)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    batch = {"src_views": {"rgb": t(frame["src_images"]), "extrinsics": t(frame["src_exts"]), "intrinsics": t(frame["src_ints"])},
             "tar_views": {"extrinsics": t(frame["tar_ext"]), "intrinsics": t(frame["tar_int"])},
             "near_far": t(frame["near_far"])}
    with torch.no_grad():
        ret, mvs_depths, blend = net(batch)
        ms = net.feature_net(batch["src_views"]["rgb"].flatten(0, 1))
        ms5 = [f.unflatten(0, (1, 3)) for f in ms]
        d, rng, vrng, vol, _ = net.depth_net(batch["src_views"]["rgb"], ms5, batch["src_views"]["extrinsics"],
                                             batch["src_views"]["intrinsics"].clone(), batch["tar_views"]["extrinsics"],
                                             batch["tar_views"]["intrinsics"].clone(), batch["near_far"])
        dec_in = torch.randn(1, 27, 32, 48)
        dec_out = net.upsampler(dec_in)
    sd = {("sd." + k): (v.numpy().astype(np.float16) if v.dtype == torch.float32 else v.numpy()) for k, v in net.state_dict().items()}
    out = dict(src_images=frame["src_images"], src_exts=frame["src_exts"], src_ints=frame["src_ints"], tar_ext=frame["tar_ext"],
               tar_int=frame["tar_int"], near_far=frame["near_far"],
               rgb=ret["rgb"].numpy(), nerf_depth=ret["nerf_depth"].numpy(), mvs_depth=ret["mvs_depth"].numpy(),
               opacity=ret["opacity"].numpy(), feat_l1=ms[1].numpy(), depth_range=rng[-1].numpy(), vol_range=vrng[-1].numpy(),
               feat_volume=vol[-1].numpy(), mvs_depth0=d[0].numpy(), dec_in=dec_in.numpy(), dec_out=dec_out.numpy(), **sd)
    path = os.path.join(HERE, "F7_network.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB; keys in state dict:", len(sd))
    more(net, RefNetwork)


def _prepared(RefNetwork, yaml, ref_sd):
    """The reference's Network under another YAML with F7's weights (same architecture: the state dicts must agree key by key)."""
    torch.manual_seed(0)
    net = RefNetwork(make_cfg(yaml)).eval()
    assert [(k, tuple(v.shape)) for k, v in net.state_dict().items()] == [(k, tuple(v.shape)) for k, v in ref_sd.items()], yaml
    net.load_state_dict(ref_sd, strict=True)
    return net


def more(net7, RefNetwork):
    """Two more pins of the call site, weights shared with F7 so that the fixtures hold inputs and outputs only:
    F7b - `configs/nerf_eval.yaml` (reweighting: True, S_max 6 adaptive; network.py:181-182) on a NeRF-synthetic-like frame;
    F7c - `configs/dtu_eval.yaml` with batch['render_scale'] = 0.5 (network.py:125-131) on a 128x192 input rendered at 64x96."""
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    sd = {k: v.clone() for k, v in net7.state_dict().items()}
    for tag, yaml, frame, scale in (("F7b_network_nerf_eval", "configs/nerf_eval.yaml", synthetic.make_frame(64, 96, V=3, B=1, scene="nerf", seed=11), None),
                                    ("F7c_network_render_scale", "configs/dtu_eval.yaml", synthetic.make_frame(128, 192, V=3, B=1, seed=12), 0.5)):
        net = _prepared(RefNetwork, yaml, sd)
        frame["src_images"] = frame["src_images"].astype(np.float16).astype(np.float32)   # stored as halves: both sides read these values
        batch = {"src_views": {"rgb": t(frame["src_images"]), "extrinsics": t(frame["src_exts"]), "intrinsics": t(frame["src_ints"])},
                 "tar_views": {"extrinsics": t(frame["tar_ext"]), "intrinsics": t(frame["tar_int"])},
                 "near_far": t(frame["near_far"])}
        if scale is not None:
            batch["render_scale"] = torch.tensor([scale])
        with torch.no_grad():
            ret, mvs_depths, blend = net(batch)
        out = dict(src_images=frame["src_images"].astype(np.float16), src_exts=frame["src_exts"], src_ints=frame["src_ints"],
                   tar_ext=frame["tar_ext"], tar_int=frame["tar_int"], near_far=frame["near_far"],
                   rgb=ret["rgb"].numpy(), nerf_depth=ret["nerf_depth"].numpy(), mvs_depth=ret["mvs_depth"].numpy(),
                   opacity=ret["opacity"].numpy(), yaml=np.array(yaml), render_scale=np.float32(scale if scale is not None else 1.0),
                   reweighting=np.bool_(net.reweighting), max_num_samples=np.int32(net.max_num_samples))
        path = os.path.join(HERE, tag + ".npz")
        np.savez_compressed(path, **out)
        print("wrote", path, os.path.getsize(path) // 1024, "KiB; rgb", tuple(ret["rgb"].shape), "reweighting", net.reweighting,
              "S_max", net.max_num_samples, "render_scale", scale)


# The 4x4 configuration the reference's YAML comments describe (configs/dtu_pretrain.yaml:23-24,33: bundle_size 4, vol_levels [0, 0],
# vol_scales [0.125, 0.25]) with the pyramid level the bundle map then reads (feat_level 0, network.py:40-43) carrying 16 channels,
# the feature width this build's kernels are specialised for.
B4_OPTS = ["nerf.bundle_size", "4", "mvs.vol_levels", "[0, 0]", "mvs.vol_scales", "[0.125, 0.25]", "fpn.feat_dims", "[16, 16, 8]"]


def bundle4(RefNetwork):
    """F7d - `configs/dtu_eval.yaml` with bundle_size 4 (network.py:31-34 accepts any power of two): its own checkpoint (the decoder
    has two up stages, decoder_rdn.py:52-63), the reference's forward on a 64x96 frame."""
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    torch.manual_seed(4)
    net = RefNetwork(make_cfg("configs/dtu_eval.yaml", B4_OPTS)).eval()
    with torch.no_grad():
        for p in net.parameters():
            p.copy_(p.half().float())
        for m in net.modules():
            if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
                m.running_mean.uniform_(-0.1, 0.1)
                m.running_var.uniform_(0.8, 1.2)
                m.running_mean.copy_(m.running_mean.half().float())
                m.running_var.copy_(m.running_var.half().float())
    frame = synthetic.make_frame(64, 96, V=3, B=1, seed=14)
    frame["src_images"] = frame["src_images"].astype(np.float16).astype(np.float32)
    batch = {"src_views": {"rgb": t(frame["src_images"]), "extrinsics": t(frame["src_exts"]), "intrinsics": t(frame["src_ints"])},
             "tar_views": {"extrinsics": t(frame["tar_ext"]), "intrinsics": t(frame["tar_int"])}, "near_far": t(frame["near_far"])}
    with torch.no_grad():
        ret, mvs_depths, blend = net(batch)
    sd = {("sd." + k): (v.numpy().astype(np.float16) if v.dtype == torch.float32 else v.numpy()) for k, v in net.state_dict().items()}
    out = dict(src_images=frame["src_images"].astype(np.float16), src_exts=frame["src_exts"], src_ints=frame["src_ints"],
               tar_ext=frame["tar_ext"], tar_int=frame["tar_int"], near_far=frame["near_far"],
               rgb=ret["rgb"].numpy(), nerf_depth=ret["nerf_depth"].numpy(), mvs_depth=ret["mvs_depth"].numpy(), opacity=ret["opacity"].numpy(),
               opts=np.array(B4_OPTS), **sd)
    path = os.path.join(HERE, "F7d_network_bundle4.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB; rgb", tuple(ret["rgb"].shape), "bundle_size", net.b_size, "feat_level", net.feat_level)


if __name__ == "__main__":
    if "--bundle4" in sys.argv:   # F7d alone (F7 .. F7c are regenerated bit for bit by the default run)
        mg._placeholders()
        from networks.gdb_nerf.network import Network as RefNetwork
        bundle4(RefNetwork)
    else:
        main()
        from networks.gdb_nerf.network import Network as RefNetwork
        bundle4(RefNetwork)
