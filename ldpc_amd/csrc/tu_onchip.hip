// tu_onchip.hip -- libldpc_hip.so, translation unit of the kernels that keep a syndrome's messages on chip (bp_small_kernel,
// bp_wave_kernel, bp_wave_ps_kernel, bp_edge_kernel, bp_edge8_kernel), with their host side (host_onchip.h: plans, tables, decode_onchip).
#include "bp_device_common.h"
#include "bp_small_kernel.h"
#include "bp_wave_kernel.h"
#include "bp_edge_kernel.h"
#include "io_kernels.h"

#include "host_handle.h"
#include "host_onchip.h"
