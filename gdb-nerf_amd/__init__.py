"""MI355X-native GDB-NeRF hot path: depth-guided bundle sampling, multi-view fetch, radiance
MLP and alpha composite as HIP kernels for gfx950 behind a C ABI (`include/gdb_nerf_hip.h`),
plus the host-side mirror of the reference's operator / plugin surface."""
__version__ = "0.1.0"
