// tu_osd.hip -- libldpc_hip.so, translation unit of ordered-statistics decoding (osd.hpp, gf2sparse_linalg.hpp): OSD-0 / OSD-E / OSD-CS
// on the rows BP left unconverged, with its host side (host_osd.h: bposd_device).
#include "bp_device_common.h"
#include "osd_kernels.h"
#include "osd_exact_kernel.h"
#include "io_kernels.h"

#include "host_handle.h"
#include "host_osd.h"

#ifdef LDPC_HIP_OSD_CLOCKS
extern "C" int ldpc_hip_debug_osd_clocks(unsigned long long *out, int reset) {
    if (out) HIPCHK(hipMemcpyFromSymbol(out, HIP_SYMBOL(osd_phase_clocks), sizeof(unsigned long long) * 16));
    if (reset) { unsigned long long z[16] = {}; HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(osd_phase_clocks), z, sizeof z)); }
    return LDPC_HIP_OK;
}
#endif

