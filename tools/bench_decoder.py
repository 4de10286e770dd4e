"""N1: the RDN decoder on the HIP library vs the PyTorch-ROCm (MIOpen) module, DTU-eval bundle map 256x320 (output 512x640),
random weights.  Mean over 50 calls after 10 warm-up calls, device-synchronised."""
import json, os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import gdb_nerf_amd  # noqa
from gdb_nerf_amd import synthetic
from gdb_nerf_amd.engine import HotPathEngine
from gdb_nerf_amd.networks.gdb_nerf.decoder_rdn import Decoder
H, W = 256, 320
torch.manual_seed(0)
dec = Decoder(27, 3, num_feats=64, num_layers=3, upscale_factor=2).cuda().eval()
frame = synthetic.make_frame(2 * H, 2 * W, V=2, seed=1)
eng = HotPathEngine(); eng.prepare({k: torch.from_numpy(v).cuda() for k, v in frame.items()})
eng.load_decoder_weights({k: v.detach() for k, v in dec.state_dict().items()}, 3)
x = torch.randn(1, 27, H, W, device="cuda")
bf = torch.zeros((H * W, 39), device="cuda"); bf[:, 12:] = x.permute(0, 2, 3, 1).reshape(H * W, 27)
def t(fn, n=50):
    for _ in range(10): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
with torch.no_grad():
    ms_torch = t(lambda: dec(x))
    ms_torch_perm = t(lambda: dec(bf.view(1, H, W, 39).permute(0, 3, 1, 2)[:, 12:]))   # as Network.forward feeds it (strided view)
ms_hip = t(lambda: eng.decode(bf, precision=1))
ms_hip_x = t(lambda: eng.decode(bf, precision=2))
with torch.no_grad():
    want = dec(x)
err = {p: float((eng.decode(bf, precision=p) - want).abs().max()) for p in (1, 2)}
flop = 2 * H * W * 9 * (27 * 64 + 3 * (64 * 32 + 96 * 32 + 128 * 64) + 64 * 256) + 2 * 4 * H * W * 64 * 3
flop_hip = 2 * H * W * 9 * (27 * 64 + 3 * (64 * 32 + 96 * 32 + 128 * 64) + 64 * 12)
print(json.dumps({"decoder_ms_torch_miopen": ms_torch, "decoder_ms_torch_from_bundle_rows": ms_torch_perm, "decoder_ms_hip": ms_hip,
                  "reference_GFLOP": flop / 1e9, "hip_GFLOP_after_folding_the_up_stage": flop_hip / 1e9,
                  "hip_TFLOPs_fp32_mfma": flop_hip / ms_hip / 1e9, "frac_of_157.3": flop_hip / ms_hip / 1e9 / 157.3,
                  "decoder_ms_hip_split_f16": ms_hip_x, "split_f16_TFLOPs_algorithmic": flop_hip / ms_hip_x / 1e9,
                  "max_abs_err_vs_torch": {"f32": err[1], "f32x": err[2]}, "output_scale": float(want.abs().max()),
                  "rows_per_wave_env": os.environ.get("GDB_DEC_R")}))
