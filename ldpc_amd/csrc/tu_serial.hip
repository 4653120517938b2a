// tu_serial.hip -- libldpc_hip.so, translation unit of the serial schedules (bp.hpp:451-545: fixed order, random order, serial_relative)
// and of soft-syndrome decoding (bp.hpp:547-660), with their host side (host_serial.h).
#include "bp_device_common.h"
#include "bp_serial_kernels.h"
#include "bp_serial_stream_kernel.h"
#include "bp_serial_var_kernel.h"
#include "bp_relative_kernel.h"
#include "bp_relative_lds_kernel.h"
#include "io_kernels.h"

#include "host_handle.h"
#include "host_serial.h"

#ifdef LDPC_SER_PROF
extern "C" int ldpc_hip_debug_serial_stream_clocks(unsigned long long *out, int reset) {
    if (out) HIPCHK(hipMemcpyFromSymbol(out, HIP_SYMBOL(ser_phase_clocks), sizeof(unsigned long long) * 8));
    if (out && reset == 2) HIPCHK(hipMemcpyFromSymbol(out + 8, HIP_SYMBOL(ser_wg_trace), sizeof(unsigned long long) * 4096 * 4));
    if (reset == 1) { unsigned long long z[8] = {}; HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(ser_phase_clocks), z, sizeof z)); }
    return LDPC_HIP_OK;
}
#endif
