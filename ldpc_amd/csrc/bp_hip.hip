// bp_hip.hip -- libldpc_hip.so: batched flooding belief propagation for gfx950 (MI355X, CDNA4).
//
// What it replaces (reference = quantumgizmos/ldpc @ /root/reference):
//   ldpc::bp::BpDecoder::bp_decode_parallel          src_cpp/bp.hpp:192-325
//   ldpc::bp::BpDecoder::initialise_log_domain_bp    src_cpp/bp.hpp:147-157
//   ldpc::gf2sparse::GF2Sparse::mulvec               src_cpp/gf2sparse.hpp:177-214
// for a BATCH of independent syndromes.  This is not a translation of that code: the reference walks
// a doubly linked list per syndrome on one CPU thread; here
//
//   * the unit of data parallelism is the syndrome.  64 syndromes form a TILE; lane l of every
//     wavefront owns syndrome l of its tile, so each per-edge message access of a wavefront is one
//     fully coalesced 512-byte row  msg[tile][edge][0..63]  (batch-minor layout),
//   * one workgroup owns one tile for the WHOLE decode (all iterations): its wavefronts stride over
//     the checks (check pass) and then over the bits (bit pass) of that tile, separated by
//     workgroup barriers only -- tiles never talk to each other, so there is no grid-wide sync,
//     no atomics and one kernel launch per decode,
//   * every lane walks its node's <= DR (row) / <= DC (column) edges SEQUENTIALLY in ascending
//     column / row order, i.e. in the reference's linked-list order (sparse_matrix_base.hpp:423-482
//     keeps both lists sorted), so every floating-point operation is performed in the reference's
//     association order -- min-sum is bit-identical, and so is product-sum with the libm-exact routines of bp_math.h,
//   * hard decisions are kept bit-packed per (tile, bit) as the wavefront's ballot; the syndrome
//     test of bp.hpp:300-302 is an XOR-gather of those 64-bit words per check,
//   * a syndrome that converges freezes its outputs (bp.hpp:300-308 early return); a tile whose 64
//     syndromes are all done retires its workgroup.
//
// No MFMA: the path is a sparse gather/scatter bound by HBM bandwidth (4 * nnz * 8 bytes per
// syndrome-iteration) and by FP64 transcendentals.
//
// Built with: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared
//
// Five translation units, compiled side by side and linked into one library (Makefile): this one carries the C ABI (handle, setup,
// decode entry points, layout kernels); tu_stream.hip, tu_serial.hip, tu_onchip.hip and tu_osd.hip carry one kernel family each
// together with its host side (host_stream.h, host_serial.h, host_onchip.h, host_osd.h).  What they call in each other is
// declared at the end of host_handle.h.  Device code lives in the kernel headers:
//   bp_device_common.h   argument blocks, buffer-descriptor message addressing, per-node arithmetic, LDS-DMA helpers
//   bp_math.h            tanh / log / division: bit-identical twins of the host libm + the fast variants
//   bp_stream_kernel.h   bp_decode_kernel        persistent workgroup per 64-syndrome tile (register / LDS-ring variants)
//   bp_spread_kernels.h  bp_spread_*_kernel      one launch per pass, a tile spread over the chip (small batches, stragglers)
//   bp_small_kernel.h    bp_small_kernel         messages resident in LDS (surface / bivariate-bicycle sized codes), slots per workgroup
//   bp_wave_kernel.h     bp_wave_kernel, bp_wave_ps_kernel   same regime, bounded degrees: one wavefront per syndrome, no workgroup barriers
//   bp_edge_kernel.h     bp_edge_kernel                      min-sum, rows <= 4 / columns <= 2 (surface-code family): lane = edge, messages in registers
//                        (lane = node; for product-sum lane = entry)
//   bp_serial_kernels.h  bp_serial_kernel, bp_softinfo_kernel   serial schedule, soft-syndrome serial min-sum
//   osd_kernels.h        osd0[_reg]_kernel, osdw[_reg]_kernel, osd_big_kernel   OSD-0 / OSD-E / OSD-CS post-processing
//   io_kernels.h         pack / unpack / transpose, H v, b8 shot data, synthetic BSC shots
//   multi_device.h       ldpc_hip_bp_multi_*: a batch sharded over several GPUs inside one process (host code only)

#include "bp_device_common.h"
#include "io_kernels.h"

#include "host_handle.h"
#include "host_setup.h"
#include "host_decode_abi.h"

#include "multi_device.h"  // ldpc_hip_bp_multi_*: one decoder over several GPUs in one process (host code over the entry points above)
