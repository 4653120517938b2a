// io_kernels.h -- layout changes either side of the decoders: pack / unpack / transpose, H v, b8 shot data, synthetic shots
// Part of libldpc_hip.so.  Included by every translation unit (bp_hip.hip, tu_*.hip): the kernels here are small and have internal
// linkage (LDPC_IO_KERNEL), so each unit that launches one carries its own copy.
#pragma once

#include "bp_device_common.h"

#define LDPC_IO_KERNEL static __global__

// One-dimensional element-wise kernels run grid-stride loops under a capped grid (flat_grid in bp_hip.hip): item counts
// such as batch * n pass 2^32 for large batches of large codes.

// Rows known to the device only (the second pass of a compacted decode, decode_stream_repacked): `count_dev` (if not null) holds the
// number of rows and overrides the by-value count; `row_map` (if not null) says which row of the CALLER's arrays row r of the launch
// is.  The host sizes grid.y from an estimate, so these kernels loop over the tiles that exist.
__device__ __forceinline__ int64_t rows_of_launch(int64_t batch, const unsigned *count_dev) {
    return count_dev ? (int64_t)__hip_atomic_load(count_dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : batch;
}

// syndromes [batch][m] u8  ->  par / nzm [tiles][m] u64, invalid [tiles] u64 (pre-zeroed)
LDPC_IO_KERNEL void pack_syndromes_kernel(const uint8_t *__restrict__ synd, int64_t batch_arg, int m,
                                      uint64_t *par, uint64_t *nzm, uint64_t *invalid,
                                      const int32_t *__restrict__ row_map = nullptr, const unsigned *count_dev = nullptr) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const int64_t batch = rows_of_launch(batch_arg, count_dev);
    for (int64_t tile = blockIdx.y; tile * LDPC_WAVE < batch; tile += gridDim.y) {
        uint64_t p = 0, z = 0, inv = 0;
        const int64_t b0 = tile * LDPC_WAVE;
        for (int l = 0; l < LDPC_WAVE; ++l) {
            const int64_t b = b0 + l;
            if (b < batch) {
                const uint8_t v = synd[(row_map ? (int64_t)row_map[b] : b) * m + i];
                p |= (uint64_t)(v & 1u) << l;
                z |= (uint64_t)(v != 0u) << l;
                inv |= (uint64_t)(v > 1u) << l;
            }
        }
        par[tile * m + i] = p;
        nzm[tile * m + i] = z;
        if (inv) atomicOr((unsigned long long *)&invalid[tile], (unsigned long long)inv);
    }
}

// dec [tiles][n] u64 -> decoding [batch][n] u8
LDPC_IO_KERNEL void unpack_decoding_kernel(const uint64_t *__restrict__ dec, int64_t batch_arg, int n,
                                       uint8_t *out, const int32_t *__restrict__ row_map = nullptr, const unsigned *count_dev = nullptr) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const int64_t batch = rows_of_launch(batch_arg, count_dev);
    for (int64_t tile = blockIdx.y; tile * LDPC_WAVE < batch; tile += gridDim.y) {
        const uint64_t v = dec[tile * n + j];
        const int64_t b0 = tile * LDPC_WAVE;
        for (int l = 0; l < LDPC_WAVE; ++l) {
            const int64_t b = b0 + l;
            if (b < batch) out[(row_map ? (int64_t)row_map[b] : b) * n + j] = (uint8_t)((v >> l) & 1ull);
        }
    }
}

// llr_t [tiles][n][64] f64 -> llr [batch][n] f64, 64x64 tiles through LDS
LDPC_IO_KERNEL void __launch_bounds__(256) transpose_llr_kernel(const double *__restrict__ llr_t,
                                                            int64_t batch_arg, int n, double *out,
                                                            const int32_t *__restrict__ row_map = nullptr, const unsigned *count_dev = nullptr) {
    __shared__ double tilebuf[LDPC_WAVE][LDPC_WAVE + 1];
    const int j0 = blockIdx.x * LDPC_WAVE;
    const int lo = threadIdx.x & 63, hi = threadIdx.x >> 6;
    const int64_t batch = rows_of_launch(batch_arg, count_dev);
    for (int64_t tile = blockIdx.y; tile * LDPC_WAVE < batch; tile += gridDim.y) {
        for (int r = 0; r < 16; ++r) {
            const int jj = r * 4 + hi;
            if (j0 + jj < n) tilebuf[jj][lo] = llr_t[((size_t)tile * n + j0 + jj) * LDPC_WAVE + lo];
        }
        __syncthreads();
        for (int r = 0; r < 16; ++r) {
            const int l = r * 4 + hi;
            const int64_t b = tile * LDPC_WAVE + l;
            if (b < batch && j0 + lo < n) out[(size_t)(row_map ? (int64_t)row_map[b] : b) * n + j0 + lo] = tilebuf[lo][l];
        }
        __syncthreads();  // (the buffer is refilled for the next tile)
    }
}

// GF2Sparse::mulvec over a batch (gf2sparse.hpp:177-214): one thread per (vector, check)
LDPC_IO_KERNEL void gf2_mulvec_kernel(const int32_t *__restrict__ row_ptr,
                                  const int32_t *__restrict__ col_idx, int m, int n,
                                  const uint8_t *__restrict__ in, int64_t batch, uint8_t *out) {
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < batch * m; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = t / m;
        const int i = (int)(t - b * m);
        uint8_t s = 0;
        for (int e = row_ptr[i]; e < row_ptr[i + 1]; ++e) s ^= in[b * n + col_idx[e]];
        out[t] = s;
    }
}

// ---- bit-packed shot data ("b8": bit i of a shot is bit i % 8 of its byte i / 8; every shot starts on a byte) --
// the wire format of the reference's sinter decoders (sinter_decoders/sinter_bposd_decoder.py:57-130)
LDPC_IO_KERNEL void unpack_b8_kernel(const uint8_t *__restrict__ in, int64_t batch, int bits, uint8_t *__restrict__ out) {
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < batch * bits; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = t / bits;
        const int i = (int)(t - b * bits);
        out[t] = (in[b * ((bits + 7) >> 3) + (i >> 3)] >> (i & 7)) & 1;
    }
}

LDPC_IO_KERNEL void pack_b8_kernel(const uint8_t *__restrict__ in, int64_t batch, int bits, uint8_t *__restrict__ out) {
    const int nb = (bits + 7) >> 3;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < batch * nb; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = t / nb;
        const int byte = (int)(t - b * nb);
        uint8_t v = 0;
        for (int q = 0; q < 8 && byte * 8 + q < bits; ++q) v |= (uint8_t)((in[b * bits + byte * 8 + q] & 1) << q);
        out[t] = v;
    }
}

// BpDecoder.decode / BpOsdDecoder.decode return the zero vector for an all-zero input without running BP
// (_bp_decoder.pyx:679-681, _bposd_decoder.pyx:118-123): converge = True, iterations reported as 0 by the batch API
LDPC_IO_KERNEL void zero_shot_shortcut_kernel(const uint8_t *__restrict__ dets_b8, int64_t batch, int m, int n, uint8_t *dec,
                                          int32_t *iters, uint8_t *conv) {
    for (int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; b < batch; b += (int64_t)gridDim.x * blockDim.x) {
        const int mb = (m + 7) >> 3;
        uint8_t any = 0;
        for (int q = 0; q < mb; ++q) {
            uint8_t v = dets_b8[b * mb + q];
            if (q == mb - 1 && (m & 7)) v &= (uint8_t)((1u << (m & 7)) - 1u);  // padding bits carry no data
            any |= v;
        }
        if (any) continue;
        for (int j = 0; j < n; ++j) dec[b * n + j] = 0;
        if (iters) iters[b] = 0;
        if (conv) conv[b] = 1;
    }
}

// predicted observables L x (mod 2) of every decoding, bit-packed: one thread per (shot, output byte)
// (SinterBpOsdDecoder.decode: `(observables_matrix @ corr) % 2`, sinter_bposd_decoder.py:128-130)
LDPC_IO_KERNEL void observables_b8_kernel(const int32_t *__restrict__ row_ptr, const int32_t *__restrict__ col_idx, int k, int n,
                                      const uint8_t *__restrict__ dec, int64_t batch, uint8_t *__restrict__ out) {
    const int nb = (k + 7) >> 3;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < batch * nb; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = t / nb;
        const int byte = (int)(t - b * nb);
        uint8_t v = 0;
        for (int q = 0; q < 8 && byte * 8 + q < k; ++q) {
            const int o = byte * 8 + q;
            uint8_t s = 0;
            for (int e = row_ptr[o]; e < row_ptr[o + 1]; ++e) s ^= dec[b * n + col_idx[e]];
            v |= (uint8_t)((s & 1) << q);
        }
        out[t] = v;
    }
}

// synthetic BSC shots: syndrome[b][i] = XOR_{j in row i} bernoulli(seed, (shot0+b)*n + j)
LDPC_IO_KERNEL void gen_bsc_syndromes_kernel(const int32_t *__restrict__ row_ptr,
                                         const int32_t *__restrict__ col_idx, int m, int n,
                                         uint64_t seed, uint64_t threshold, int64_t shot0,
                                         int64_t batch, uint8_t *synd) {
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < batch * m; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = t / m;
        const int i = (int)(t - b * m);
        const uint64_t base = (uint64_t)(shot0 + b) * (uint64_t)n;
        uint8_t s = 0;
        for (int e = row_ptr[i]; e < row_ptr[i + 1]; ++e)
            s ^= (uint8_t)((sm64(seed, base + (uint64_t)col_idx[e]) >> 11) < threshold);
        synd[t] = s;
    }
}

LDPC_IO_KERNEL void gen_bsc_errors_kernel(int n, uint64_t seed, uint64_t threshold, int64_t shot0,
                                      int64_t batch, uint8_t *err) {
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < batch * n; t += (int64_t)gridDim.x * blockDim.x) {
        const uint64_t idx = (uint64_t)shot0 * (uint64_t)n + (uint64_t)t;
        err[t] = (uint8_t)((sm64(seed, idx) >> 11) < threshold);
    }
}


// ---- row gather / scatter by an index list (repacking of the syndromes a first short pass left unconverged) ----
template <class T>
__global__ void gather_rows_kernel(const T *__restrict__ src, const int32_t *__restrict__ list, int64_t count, int width,
                                   T *__restrict__ dst) {
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < count * width; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = t / width;
        dst[t] = src[(int64_t)list[r] * width + (t - r * width)];
    }
}
// message state of listed syndromes, lane by lane, out of the 64-syndrome tiles of a first pass into dense tiles: syndrome
// list[r] (tile list[r] / 64, lane list[r] % 64) becomes lane r % 64 of tile r / 64; src, dst: [tiles][nnz][64] doubles.
// One wavefront per (destination tile, edge): 64 gathered 8-byte loads (the live lanes of a source row share sectors), one 512-byte store.
LDPC_IO_KERNEL void __launch_bounds__(256) gather_lane_state_kernel(const double *__restrict__ src, const int32_t *__restrict__ list, int64_t count_arg,
                                                                int nnz, int edges_per_wave, double *__restrict__ dst, const unsigned *count_dev = nullptr) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t count = rows_of_launch(count_arg, count_dev);
    for (int64_t tile = blockIdx.y; tile * 64 < count; tile += gridDim.y) {
        const int64_t r = tile * 64 + lane;
        const bool live = r < count;
        const int64_t b = live ? (int64_t)list[r] : 0;
        const double *from = src + ((b >> 6) * (int64_t)nnz) * 64 + (b & 63);
        double *to = dst + (tile * (int64_t)nnz) * 64 + lane;
        const int e0 = (blockIdx.x * 4 + wave) * edges_per_wave;
        for (int e = e0; e < e0 + edges_per_wave && e < nnz; ++e) to[(int64_t)e * 64] = live ? from[(int64_t)e * 64] : 0.0;
    }
}

template <class T>
__global__ void scatter_rows_kernel(const T *__restrict__ src, const int32_t *__restrict__ list, int64_t count, int width,
                                    T *__restrict__ dst) {
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < count * width; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = t / width;
        dst[(int64_t)list[r] * width + (t - r * width)] = src[t];
    }
}

// How many rows converged after exactly j iterations (bin j, j capped at 255) and how many did not converge (bin 0):
// what the host needs to see whether a short first pass + a second pass over the rest would have been cheaper
LDPC_IO_KERNEL void __launch_bounds__(256) iteration_histogram_kernel(const int32_t *__restrict__ iters, const uint8_t *__restrict__ conv,
                                                                  int64_t batch, unsigned *__restrict__ hist) {
    __shared__ unsigned local[256];
    local[threadIdx.x] = 0;
    __syncthreads();
    for (int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; b < batch; b += (int64_t)gridDim.x * blockDim.x) {
        const int it = iters[b];
        atomicAdd(&local[conv[b] ? (it < 1 ? 1 : it > 255 ? 255 : it) : 0], 1u);
    }
    __syncthreads();
    if (local[threadIdx.x]) atomicAdd(&hist[threadIdx.x], local[threadIdx.x]);
}

// Rows that need OSD are a few percent of a batch and scattered: list them first, then persistent wavefronts pull rows
// from the list, so every resident wavefront has work (one wavefront per batch row would leave the chip almost empty).
// `status` (or nullptr): the OSD status array of the batch, cleared here (0 = BP converged, OSD not run) so that BP + OSD needs no fill of its own.
LDPC_IO_KERNEL void __launch_bounds__(256) osd_collect_kernel(const uint8_t *__restrict__ conv, int64_t batch, int32_t *list, unsigned *counters,
                                                              uint8_t *status = nullptr) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool need = b < batch && !conv[b];
    if (status && b < batch) status[b] = 0;
    const uint64_t mask = __ballot(need);
    if (!mask) return;
    const int lane = threadIdx.x & 63;
    unsigned base = 0;
    if (lane == 0) base = atomicAdd(&counters[0], (unsigned)__builtin_popcountll(mask));
    base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
    if (need) list[base + __builtin_popcountll(mask & ((1ull << lane) - 1ull))] = (int32_t)b;
}


// ---- copy-rate probe (ldpc_hip_bp_copy_probe): what HBM gives THIS box, now, for the traffic pattern of the streamed kernels --------
// A workgroup per tile copies the tile's `nseg` 512-byte segments (64 lanes x one double: one edge of the code) from src to dst, six
// segments in flight per wavefront, non-temporal -- the streamed kernels' access size and policy with no arithmetic
// (tools/membench/segcopy.hip is the stand-alone form with gather / scatter orders; every order copies at the same rate there).
LDPC_IO_KERNEL void __launch_bounds__(768) segcopy_probe_kernel(const double *__restrict__ src, double *__restrict__ dst, int nseg) {
    constexpr int U = 6;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), nwaves = (int)(blockDim.x >> 6);
    const size_t base = (size_t)blockIdx.x * (size_t)nseg * 64;
    for (int e0 = wave * U; e0 < nseg; e0 += nwaves * U) {
        double v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e = e0 + u < nseg ? e0 + u : nseg - 1;
            v[u] = __builtin_nontemporal_load(src + base + (size_t)e * 64 + lane);
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (e0 + u < nseg) __builtin_nontemporal_store(v[u], dst + base + (size_t)(e0 + u) * 64 + lane);
    }
}
