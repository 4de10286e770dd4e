"""The HIP decoder alone, N calls at the DTU-eval bundle map (256x320), for rocprofv3 (kernel trace / PMC passes).
usage: run_decoder.py [n_calls] [precision 1|2]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import gdb_nerf_amd  # noqa
from gdb_nerf_amd import synthetic
from gdb_nerf_amd.engine import HotPathEngine
from gdb_nerf_amd.networks.gdb_nerf.decoder_rdn import Decoder
n, prec = int(sys.argv[1]) if len(sys.argv) > 1 else 10, int(sys.argv[2]) if len(sys.argv) > 2 else 1
H, W = 256, 320
torch.manual_seed(0)
dec = Decoder(27, 3, num_feats=64, num_layers=3, upscale_factor=2).cuda().eval()
frame = synthetic.make_frame(2 * H, 2 * W, V=2, seed=1)
eng = HotPathEngine(); eng.prepare({k: torch.from_numpy(v).cuda() for k, v in frame.items()})
eng.load_decoder_weights({k: v.detach() for k, v in dec.state_dict().items()}, 3)
bf = torch.zeros((H * W, 39), device="cuda"); bf[:, 12:] = torch.randn(H * W, 27, device="cuda")
for _ in range(3 * n):   # warm-up (clocks)
    eng.decode(bf, precision=prec)
torch.cuda.synchronize()
for _ in range(n):
    eng.decode(bf, precision=prec)
torch.cuda.synchronize()
