// Yardstick kernels for the HBM-bound passes (development aid, built on the GPU box by tools/gpu_stream_bw.py):
// what this box's HBM gives a hand-written 16 B/lane stream, for read+write mixes like the pool+statistics pass.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

// grid-stride copy, 16 B per lane per access, UNROLL independent loads in flight per lane
template <int UNROLL, bool NT>
__global__ __launch_bounds__(256) void copy_kernel(const u32x4 *__restrict__ in, u32x4 *__restrict__ out, size_t n) {
  const size_t stride = (size_t)gridDim.x * 256 * UNROLL;
  for (size_t i = (size_t)blockIdx.x * 256 * UNROLL + threadIdx.x; i < n; i += stride) {
    u32x4 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u)
      if (i + u * 256 < n) v[u] = NT ? __builtin_nontemporal_load(in + i + u * 256) : in[i + u * 256];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u)
      if (i + u * 256 < n) {
        if (NT) __builtin_nontemporal_store(v[u], out + i + u * 256);
        else out[i + u * 256] = v[u];
      }
  }
}
// read 4 units, write 1 (the pool pass's ratio: the tap read once, a quarter written).
// Round 3's form of this kernel kept ONE 16-byte load per lane in flight on each of four streams 512 MiB apart and
// measured BELOW the 1:1 copy of this file -- a kernel artefact, not a hardware ceiling.  This form
// walks CONTIGUOUS input: a block takes a run of 4*UNROLL KiB-sized rows (256 lanes x 16 B), issues all 4*UNROLL loads
// before any use (UNROLL = 4: sixteen independent 16-byte loads per lane in flight, as many as copy_kernel<4> has plus
// its stores), folds each group of four consecutive rows into one and writes UNROLL rows.  The grid is sized to the chip
// by the caller (blocks = CUs x k) and strides over the buffer.
template <int UNROLL, bool NT>
__global__ __launch_bounds__(256) void read4_write1_kernel(const u32x4 *__restrict__ in, u32x4 *__restrict__ out, size_t n_out) {
  const size_t out_per_iter = (size_t)256 * UNROLL;  // output vectors a block produces per iteration
  const size_t stride = (size_t)gridDim.x * out_per_iter;
  for (size_t o0 = (size_t)blockIdx.x * out_per_iter; o0 < n_out; o0 += stride) {
    const u32x4 *src = in + 4 * o0 + threadIdx.x;
    u32x4 v[4 * UNROLL];
#pragma unroll
    for (int u = 0; u < 4 * UNROLL; ++u) {
      const size_t o = o0 + (size_t)(u >> 2) * 256 + threadIdx.x;
      if (o < n_out) v[u] = NT ? __builtin_nontemporal_load(src + u * 256) : src[u * 256];
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const size_t o = o0 + (size_t)u * 256 + threadIdx.x;
      if (o < n_out) {
        const u32x4 r = v[4 * u] ^ v[4 * u + 1] ^ v[4 * u + 2] ^ v[4 * u + 3];
        if (NT) __builtin_nontemporal_store(r, out + o);
        else out[o] = r;
      }
    }
  }
}
// write only (fill), for the producer side of the cache-residency probe
template <int UNROLL>
__global__ __launch_bounds__(256) void fill_kernel(u32x4 *__restrict__ out, size_t n, uint32_t seed) {
  const size_t stride = (size_t)gridDim.x * 256 * UNROLL;
  for (size_t i = (size_t)blockIdx.x * 256 * UNROLL + threadIdx.x; i < n; i += stride) {
#pragma unroll
    for (int u = 0; u < UNROLL; ++u)
      if (i + u * 256 < n) out[i + u * 256] = (u32x4){seed, (uint32_t)i, (uint32_t)u, seed ^ (uint32_t)i};
  }
}
template <int UNROLL, bool NT>
__global__ __launch_bounds__(256) void read_kernel(const u32x4 *__restrict__ in, u32x4 *__restrict__ out, size_t n) {
  const size_t stride = (size_t)gridDim.x * 256 * UNROLL;
  u32x4 acc = {0, 0, 0, 0};
  for (size_t i = (size_t)blockIdx.x * 256 * UNROLL + threadIdx.x; i < n; i += stride) {
#pragma unroll
    for (int u = 0; u < UNROLL; ++u)
      if (i + u * 256 < n) acc ^= NT ? __builtin_nontemporal_load(in + i + u * 256) : in[i + u * 256];
  }
  if (acc[0] == 0x12345678u && acc[1] == 0x9abcdef0u) out[threadIdx.x] = acc;  // (never true on the test data: keeps the loads)
}

// kind: 0 copy, 1 read 4 : write 1 (unroll = independent row groups per lane: 1, 2 or 4), 2 read only, 3 fill
extern "C" int stream_run2(int kind, int nt, int unroll, const void *in, void *out, size_t bytes, int blocks, void *stream) {
  hipStream_t st = (hipStream_t)stream;
  const size_t n = bytes / 16;
  const u32x4 *i = (const u32x4 *)in;
  u32x4 *o = (u32x4 *)out;
  if (kind == 0) { if (nt) copy_kernel<4, true><<<blocks, 256, 0, st>>>(i, o, n); else copy_kernel<4, false><<<blocks, 256, 0, st>>>(i, o, n); }
  else if (kind == 1) {
#define R4W1(U) { if (nt) read4_write1_kernel<U, true><<<blocks, 256, 0, st>>>(i, o, n / 4); else read4_write1_kernel<U, false><<<blocks, 256, 0, st>>>(i, o, n / 4); }
    if (unroll >= 4) R4W1(4) else if (unroll == 2) R4W1(2) else R4W1(1)
#undef R4W1
  }
  else if (kind == 2) { if (nt) read_kernel<4, true><<<blocks, 256, 0, st>>>(i, o, n); else read_kernel<4, false><<<blocks, 256, 0, st>>>(i, o, n); }
  else fill_kernel<4><<<blocks, 256, 0, st>>>(o, n, (uint32_t)nt);
  return (int)hipGetLastError();
}
extern "C" int stream_run(int kind, int nt, const void *in, void *out, size_t bytes, int blocks, void *stream) {
  return stream_run2(kind, nt, 4, in, out, bytes, blocks, stream);
}
