// tu_stream.hip -- libldpc_hip.so, translation unit of the streamed kernels: bp_decode_kernel (persistent workgroup per 64-syndrome
// tile, bp.hpp:192-325) and the chip-wide per-pass kernels bp_spread_*, with their host side (host_stream.h: decode_device, the
// two-pass decode with lane compaction).  See bp_hip.hip for the design notes and the list of kernel headers.
#include "bp_device_common.h"
#include "bp_stream_kernel.h"
#include "bp_spread_kernels.h"
#include "bp_flood_lane_kernel.h"
#include "io_kernels.h"

#include "host_handle.h"

typedef void (*bp_kernel_t)(const BpArgs);
typedef void (*spread_kernel_t)(const SpreadArgs);

// (the LOOP forms serve the slots beyond the first 32 of a compacted list: bp_spread_kernels.h, "late rounds")
template <int METHOD, int MATH, bool LOOP>
static void pick_spread_m(int max_row, int max_col, bool nt, spread_kernel_t &kc, spread_kernel_t &kb) {
    if (nt) {
        kc = max_row <= 8 ? bp_spread_check_kernel<METHOD, MATH, 8, 1, LOOP> : bp_spread_check_kernel<METHOD, MATH, 16, 1, LOOP>;
        kb = max_col <= 4 ? bp_spread_bit_kernel<METHOD, MATH, 4, 1, LOOP> : bp_spread_bit_kernel<METHOD, MATH, 8, 1, LOOP>;
    } else {
        kc = max_row <= 8 ? bp_spread_check_kernel<METHOD, MATH, 8, 0, LOOP> : bp_spread_check_kernel<METHOD, MATH, 16, 0, LOOP>;
        kb = max_col <= 4 ? bp_spread_bit_kernel<METHOD, MATH, 4, 0, LOOP> : bp_spread_bit_kernel<METHOD, MATH, 8, 0, LOOP>;
    }
}

struct KernelChoice {
    bp_kernel_t fn;
    int ring_slot_bytes;  // 0: register-prefetch variant, no dynamic LDS
    int ring_depth;
    int max_waves = 16;   // wavefronts per workgroup the variant was compiled for (stream_max_waves)
    bool var_ring = false;  // the variable-degree ring (LDPC_RING_VAR): dynamic LDS = units x 1 KiB per wavefront, chosen at launch
};

template <int METHOD, int MATH>
static KernelChoice pick_kernel(int max_row, int max_col, int ring_depth, bool var_ring) {
    if (var_ring) return {bp_decode_kernel<METHOD, MATH, 16, 8, LDPC_RING_VAR>, 0, 0, stream_max_waves(16, LDPC_RING_VAR), true};
    // Register arrays are sized by the template bounds, so the common regular codes get exact fits:
    // (3,6)-LDPC / bivariate-bicycle rows of 6 and columns of 3 use the LDS-DMA ring variant.
    if (ring_depth == 2 && max_row == 6 && max_col == 3) return {bp_decode_kernel<METHOD, MATH, 6, 3, 2>, 3 * 1024, 2};
    if (ring_depth >= 3 && max_row == 6 && max_col == 3) return {bp_decode_kernel<METHOD, MATH, 6, 3, 3>, 3 * 1024, 3};
    // (8,4): min-sum only by default -- product-sum runs 3 % faster on the per-pass kernels from the first iteration (0.605 against 0.587 of HBM at
    // 512 tiles, profiles/r6_ring_shapes.jsonl); ldpc_hip_bp_set_ring(h, 3) still selects the ring for it
    if (ring_depth >= (METHOD == LDPC_HIP_PRODUCT_SUM ? 3 : 2) && max_row == 8 && max_col == 4) return {bp_decode_kernel<METHOD, MATH, 8, 4, 3>, 4 * 1024, 3};
    // round 6: (3,4)- and (3,5)-regular codes (rows of 4 / 5 entries, columns of 3), product-sum only: +4 % over the per-pass kernels there,
    // nothing (3,4) or -2.5 % (3,5) for min-sum against the register variant; a (10,5) ring measured equal (product-sum) and -8 % (min-sum)
    // and is not built (profiles/r6_ring_shapes.jsonl)
    if (METHOD == LDPC_HIP_PRODUCT_SUM && ring_depth >= 2 && max_row == 4 && max_col == 3) return {bp_decode_kernel<METHOD, MATH, 4, 3, 2>, 3 * 1024, 2};
    if (METHOD == LDPC_HIP_PRODUCT_SUM && ring_depth >= 2 && max_row == 5 && max_col == 3) return {bp_decode_kernel<METHOD, MATH, 5, 3, 2>, 3 * 1024, 2};
    if (max_row <= 4 && max_col <= 3) return {bp_decode_kernel<METHOD, MATH, 4, 3, 0>, 0, 0};
    if (max_row <= 6 && max_col <= 3) return {bp_decode_kernel<METHOD, MATH, 6, 3, 0>, 0, 0};
    if (max_row <= 8 && max_col <= 4) return {bp_decode_kernel<METHOD, MATH, 8, 4, 0>, 0, 0, stream_max_waves(8, 0)};
    if (max_row <= 8 && max_col <= 8) return {bp_decode_kernel<METHOD, MATH, 8, 8, 0>, 0, 0, stream_max_waves(8, 0)};
    if (max_col <= 8) return {bp_decode_kernel<METHOD, MATH, 16, 8, 0>, 0, 0, stream_max_waves(16, 0)};
    return {bp_decode_kernel<METHOD, MATH, 16, 16, 0>, 0, 0, stream_max_waves(16, 0)};  // heavier nodes take the streaming path inside
}

#include "host_stream.h"
