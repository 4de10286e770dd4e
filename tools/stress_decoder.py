"""One-off stress: N seeded random bundle-map shapes / batch sizes / block counts, HIP decoder (both precisions) vs the PyTorch
module on the same GPU (same check as tests/test_decoder.py::test_hip_decoder_matches_torch_module).  usage: stress_decoder.py [N] [seed]"""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import gdb_nerf_amd  # noqa: F401
from gdb_nerf_amd import synthetic
from gdb_nerf_amd.engine import HotPathEngine
from gdb_nerf_amd.networks.gdb_nerf.decoder_rdn import Decoder
N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 3)
worst, bad = {1: 0.0, 2: 0.0}, 0
for i in range(N):
    B, H, W, layers = int(rng.integers(1, 4)), int(rng.integers(1, 150)), int(rng.integers(1, 200)), int(rng.integers(1, 4))
    torch.manual_seed(i)
    dec = Decoder(27, 3, num_feats=64, num_layers=layers, upscale_factor=2).cuda().eval()
    with torch.no_grad():
        for p in dec.parameters(): p.mul_(1.5)
    frame = synthetic.make_frame(2 * H, 2 * W, V=2, B=B, seed=1)
    eng = HotPathEngine(); eng.prepare({k: torch.from_numpy(v).cuda() for k, v in frame.items()})
    eng.load_decoder_weights({k: v.detach() for k, v in dec.state_dict().items()}, layers)
    ld = int(rng.choice([39, 41]))
    x = torch.randn(B, 27, H, W, device="cuda")
    bf = torch.zeros((B * H * W, ld), device="cuda"); bf[:, 12:39] = x.permute(0, 2, 3, 1).reshape(B * H * W, 27)
    with torch.no_grad(): want = dec(x)
    scale = max(1.0, float(want.abs().max()))
    for prec in (1, 2):
        e = float((eng.decode(bf, precision=prec) - want).abs().max()) / scale
        worst[prec] = max(worst[prec], e)
        if not e <= 3e-5:
            bad += 1; print(f"FAIL case {i} B={B} {H}x{W} layers={layers} ld={ld} precision {prec}: rel err {e:.3e}", flush=True)
print(f"done: {N} cases x 2 precisions, worst relative err fp32 {worst[1]:.3e}, split-f16 {worst[2]:.3e}, failures {bad}")
sys.exit(1 if bad else 0)
