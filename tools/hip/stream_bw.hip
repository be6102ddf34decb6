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
// read 4 units, write 1 (the pool pass's ratio: the tap read once, a quarter written)
template <int UNROLL, bool NT>
__global__ __launch_bounds__(256) void read4_write1_kernel(const u32x4 *__restrict__ in, u32x4 *__restrict__ out, size_t n_out) {
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n_out; i += stride) {
    u32x4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = NT ? __builtin_nontemporal_load(in + u * n_out + i) : in[u * n_out + i];
    u32x4 r = v[0] ^ v[1] ^ v[2] ^ v[3];
    if (NT) __builtin_nontemporal_store(r, out + i);
    else out[i] = r;
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

extern "C" int stream_run(int kind, int nt, const void *in, void *out, size_t bytes, int blocks, void *stream) {
  hipStream_t st = (hipStream_t)stream;
  const size_t n = bytes / 16;
  const u32x4 *i = (const u32x4 *)in;
  u32x4 *o = (u32x4 *)out;
  if (kind == 0) { if (nt) copy_kernel<4, true><<<blocks, 256, 0, st>>>(i, o, n); else copy_kernel<4, false><<<blocks, 256, 0, st>>>(i, o, n); }
  else if (kind == 1) { if (nt) read4_write1_kernel<1, true><<<blocks, 256, 0, st>>>(i, o, n / 4); else read4_write1_kernel<1, false><<<blocks, 256, 0, st>>>(i, o, n / 4); }
  else { if (nt) read_kernel<4, true><<<blocks, 256, 0, st>>>(i, o, n); else read_kernel<4, false><<<blocks, 256, 0, st>>>(i, o, n); }
  return (int)hipGetLastError();
}
