"""Soak of the HIP decoder (its squeeze-excitation gate is a last-arriver reduction across workgroups / XCDs): N back-to-back decodes
per precision at the DTU-eval bundle map and at a ragged batch-2 map, every 50th result compared bit for bit with the first.
usage: soak_decoder.py [N=3000]"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import gdb_nerf_amd  # noqa
from gdb_nerf_amd import synthetic
from gdb_nerf_amd.engine import HotPathEngine
from gdb_nerf_amd.networks.gdb_nerf.decoder_rdn import Decoder
N = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
bad = 0
for (B, H, W) in ((1, 256, 320), (2, 37, 70)):
    torch.manual_seed(0)
    dec = Decoder(27, 3, num_feats=64, num_layers=3, upscale_factor=2).cuda().eval()
    frame = synthetic.make_frame(2 * H, 2 * W, V=2, B=B, seed=1)
    eng = HotPathEngine(); eng.prepare({k: torch.from_numpy(v).cuda() for k, v in frame.items()})
    eng.load_decoder_weights({k: v.detach() for k, v in dec.state_dict().items()}, 3)
    bf = torch.zeros((B * H * W, 41), device="cuda"); bf[:, 12:39] = torch.randn(B * H * W, 27, device="cuda")
    for prec in (1, 2):
        ref = eng.decode(bf, precision=prec).clone()
        t0 = time.time(); mism = 0
        for i in range(N):
            out = eng.decode(bf, precision=prec)
            if i % 50 == 49 and not torch.equal(out, ref): mism += 1
        torch.cuda.synchronize()
        bad += mism
        print(f"({B},{H},{W}) precision {prec}: {N} decodes in {time.time() - t0:.1f} s, {mism} mismatching checks", flush=True)
print("decoder soak:", "FAILED" if bad else "ok, every checked result bit-identical to the first")
