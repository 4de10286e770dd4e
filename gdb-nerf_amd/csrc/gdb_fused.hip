// placeholder until the fused kernel lands
#include "gdb_internal.h"
int gdb_fail(int code, const char* fmt, ...);
size_t gdb_mfma_section_floats() { return 0; }
void gdb_pack_mfma_section(const float*, float*) {}
extern "C" int gdb_render_bundles_fused(const GdbConfig*, const GdbFrame*, const void*, const float*, int32_t, int32_t,
                                        int32_t, float*, float*, float*, void*) {
    return gdb_fail(GDB_E_BADARG, "fused kernel not built yet");
}
