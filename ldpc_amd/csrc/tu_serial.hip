// tu_serial.hip -- libldpc_hip.so, translation unit of the serial schedules (bp.hpp:451-545: fixed order, random order, serial_relative)
// and of soft-syndrome decoding (bp.hpp:547-660), with their host side (host_serial.h).
#include "bp_device_common.h"
#include "bp_serial_kernels.h"
#include "bp_serial_stream_kernel.h"
#include "bp_relative_kernel.h"
#include "bp_relative_lds_kernel.h"
#include "io_kernels.h"

#include "host_handle.h"
#include "host_serial.h"
