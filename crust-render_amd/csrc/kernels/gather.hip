// The multi-GPU path's one data-path step besides the collective itself: assembling the frame from the ranks' padded
// tile buffers. The frame's 16x16 tiles (tracer.rs:424, :1671-1686) are dealt round-robin to the ranks
// (crt_shard_pixels); after the last batch every rank holds `padded` pixels (its shard, zero-padded to the common
// length crt_shard_padded_count) and ONE gather / all-gather of those buffers (RCCL over xGMI; the host's own call —
// torch.distributed in bench.py, ncclAllGather in a native host) leaves world x padded x 3 floats on the receiving
// rank. crt_gather_plan_assemble scatters them into the frame with one launch: the index plan (which pixel slot k of
// rank r is) lives in HBM, built once per (width, height, world).
#include <hip/hip_runtime.h>

#include <new>
#include <vector>

#include "../crt_internal.h"

struct CrtGatherPlan {
  uint32_t width = 0, height = 0, world = 0;
  size_t padded = 0;
  uint32_t *d_index = nullptr;  // [world][padded]: linear pixel index j * width + i, or CRT_INVALID_ID for padding
};

namespace crt {
namespace {
__global__ __launch_bounds__(256) void k_assemble(const float *__restrict__ recv, const uint32_t *__restrict__ index, size_t n,
                                                   float *__restrict__ frame) {
  for (size_t k = (size_t)blockIdx.x * 256 + threadIdx.x; k < n; k += (size_t)gridDim.x * 256) {
    const uint32_t p = index[k];
    if (p == CRT_INVALID_ID) continue;
    frame[3 * (size_t)p] = recv[3 * k]; frame[3 * (size_t)p + 1] = recv[3 * k + 1]; frame[3 * (size_t)p + 2] = recv[3 * k + 2];
  }
}
}  // namespace
}  // namespace crt

using namespace crt;

extern "C" {

size_t crt_shard_padded_count(uint32_t width, uint32_t height, uint32_t world) {
  if (world == 0) return 0;
  const size_t tx = (width + 15) / 16, ty = (height + 15) / 16;
  const size_t tiles = (tx * ty + world - 1) / world;  // the most tiles any rank owns
  return ((tiles * 256 + 255) / 256) * 256;
}

CrtGatherPlan *crt_gather_plan_new(uint32_t width, uint32_t height, uint32_t world) {
  if (world == 0 || width == 0 || height == 0) return nullptr;
  if (!device_ok()) return nullptr;
  CrtGatherPlan *p = new (std::nothrow) CrtGatherPlan();
  if (!p) return nullptr;
  p->width = width; p->height = height; p->world = world;
  p->padded = crt_shard_padded_count(width, height, world);
  const int rc = abi_guard("crt_gather_plan_new", [&] {  // the host copy of the index plan is a std::vector
    std::vector<uint32_t> idx((size_t)world * p->padded, CRT_INVALID_ID);
    for (uint32_t r = 0; r < world; r++) crt_shard_pixels(width, height, r, world, idx.data() + (size_t)r * p->padded);
    return CRT_HIP_OK(hipMalloc(&p->d_index, idx.size() * 4)) &&
                   CRT_HIP_OK(hipMemcpy(p->d_index, idx.data(), idx.size() * 4, hipMemcpyHostToDevice))
               ? (int)CRT_OK
               : (int)CRT_ERR_NO_DEVICE;
  });
  if (rc != CRT_OK) {
    if (p->d_index) (void)hipFree(p->d_index);
    delete p;
    return nullptr;
  }
  return p;
}
void crt_gather_plan_free(CrtGatherPlan *p) {
  if (!p) return;
  if (p->d_index) (void)hipFree(p->d_index);
  delete p;
}
size_t crt_gather_plan_padded_count(const CrtGatherPlan *p) { return p ? p->padded : 0; }
int crt_gather_plan_assemble(const CrtGatherPlan *p, const float *d_recv, float *d_frame, void *stream) {
  if (!p || !d_recv || !d_frame) return CRT_ERR_BAD_ARG;
  const size_t n = (size_t)p->world * p->padded;
  size_t blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(k_assemble, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, d_recv, p->d_index, n, d_frame);
  return CRT_HIP_OK(hipGetLastError()) ? CRT_OK : CRT_ERR_NO_DEVICE;
}

}  // extern "C"
