"""Do MFMAs and fp32 VALU FMAs overlap on a gfx950 SIMD?  (tools/ubench/peaks.hip k_mix)  One workgroup per CU; with 512 threads each
SIMD hosts two waves: both MFMA, both VALU, or one of each.  If the pipes are separate, the mixed run takes max(t_mfma, t_valu) of the
one-wave-per-SIMD runs; if they share the datapath, their sum."""
import ctypes as C, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import gdb_nerf_amd  # noqa
from gdb_nerf_amd import build as _b
lib = C.CDLL(_b.build_peaks())
lib.gdb_peak_mix.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
sink = torch.zeros(4096, device="cuda")
st = torch.cuda.current_stream().cuda_stream
def t(threads, iters, mode, f16, blocks=256):
    best = 1e9
    for _ in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); lib.gdb_peak_mix(sink.data_ptr(), blocks, threads, iters, mode, f16, st); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best
iters = 20000
for f16 in (0, 1):
    name = "v_mfma_f32_32x32x16_f16" if f16 else "v_mfma_f32_32x32x2_f32"
    a1 = t(256, iters, 0, f16)   # 256 threads: one MFMA wave per SIMD
    a2 = t(512, iters, 0, f16)   # 512 threads: two MFMA waves per SIMD
    print(f"{name}: one MFMA wave per SIMD {a1:.3f} ms, two {a2:.3f} ms")
    for label, m_all, m_mix in (("scalar v_fma_f32", 1, 2), ("packed v_pk_fma_f32", 3, 4)):
        v1, v2, mix = t(256, iters, m_all, f16), t(512, iters, m_all, f16), t(512, iters, m_mix, f16)
        print(f"   {label:20s}: one VALU wave per SIMD {v1:.3f} ms, two {v2:.3f} ms; one MFMA + one VALU wave {mix:.3f} ms "
              f"(separate pipes would give max = {max(a1, v1):.3f}, a shared one the sum = {a1 + v1:.3f})")
