// The multi-GPU path from a NATIVE host: one process per GPU, the collective driven by the host itself over RCCL.
//
//   RANK=r WORLD_SIZE=n [LOCAL_RANK=d] [CRT_NCCL_ID_FILE=path] crt_rccl_host <scene.bin> <film.bin>
//
// What bench.py does with torch.distributed, here in C++ over include/crt.h and <rccl/rccl.h> (the reference
// renders its 16x16 tiles on Rayon workers into one shared buffer, tracer.rs:424-459, :1671-1686):
//   every rank commits the (replicated) scene and owns the tiles crt_shard_pixels deals it     crt_renderer_new(.., rank, world)
//   traces its samples — no collective on the data path                                        crt_render_samples
//   resolves its shard into the front of a zeroed padded x 3 float buffer                      crt_film_resolve, crt_shard_padded_count
//   ONE ncclAllGather of padded x 3 floats per rank (xGMI point-to-point underneath)
//   one launch scatters world x padded x 3 floats into the frame                               crt_gather_plan_assemble
//   the job's RayStats: one ncclAllReduce of the eight 64-bit counters
// Rank 0 writes the frame and the counters (the format of ../host_c/crt_host.c). The ncclUniqueId travels through a file
// (rank 0 writes it, the others wait for it): no MPI, no Python. With WORLD_SIZE=1 the same calls run on one GPU — that
// is the case tests/test_gpu_host_c.py can run on a one-GPU box; the frame must equal the single-process hosts'.
// CRT_RCCL_LOOPBACK=1 (tests): ONE process plays every rank in turn — each rank's renderer, its shard resolved to where
// the all-gather would have put it (recv + rank x padded x 3), counters summed on the host — and no communicator is
// made: the multi-rank data layout of this file, checked on a box where RCCL cannot take two ranks.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "../host_c/scene_file.h"

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "crt_rccl_host: %s: %s\n", #x, hipGetErrorString(e_)); return 3; } } while (0)
#define NCCL_OK(x) do { ncclResult_t e_ = (x); if (e_ != ncclSuccess) { fprintf(stderr, "crt_rccl_host: %s: %s\n", #x, ncclGetErrorString(e_)); return 3; } } while (0)

static_assert(sizeof(CrtRayStats) == 64, "eight 64-bit counters");
static int env_int(const char *name, int dflt) { const char *v = getenv(name); return v && *v ? atoi(v) : dflt; }

// rank 0 creates the id and publishes it (write to a temporary name, then rename: readers never see half a file)
static int exchange_id(ncclUniqueId *id, int rank, int world) {
  if (world == 1) return ncclGetUniqueId(id) == ncclSuccess ? 0 : 3;
  const char *path = getenv("CRT_NCCL_ID_FILE");
  if (!path || !*path) { fprintf(stderr, "crt_rccl_host: WORLD_SIZE > 1 needs CRT_NCCL_ID_FILE\n"); return 2; }
  if (rank == 0) {
    if (ncclGetUniqueId(id) != ncclSuccess) return 3;
    std::vector<char> tmp(strlen(path) + 8);
    snprintf(tmp.data(), tmp.size(), "%s.tmp", path);
    FILE *f = fopen(tmp.data(), "wb");
    if (!f || fwrite(id, sizeof *id, 1, f) != 1) { perror(path); return 2; }
    fclose(f);
    return rename(tmp.data(), path) == 0 ? 0 : 2;
  }
  for (int tries = 0; tries < 600; tries++) {  // up to a minute
    FILE *f = fopen(path, "rb");
    if (f) {
      const size_t got = fread(id, sizeof *id, 1, f);
      fclose(f);
      if (got == 1) return 0;
    }
    std::this_thread::sleep_for(std::chrono::milliseconds(100));
  }
  fprintf(stderr, "crt_rccl_host: rank %d never saw %s\n", rank, path);
  return 3;
}

int main(int argc, char **argv) {
  if (argc != 3) { fprintf(stderr, "usage: RANK=r WORLD_SIZE=n crt_rccl_host <scene.bin> <film.bin>\n"); return 2; }
  const int rank = env_int("RANK", 0), world = env_int("WORLD_SIZE", 1);
  if (world < 1 || rank < 0 || rank >= world) { fprintf(stderr, "crt_rccl_host: bad RANK / WORLD_SIZE\n"); return 2; }
  World w;
  const int rc = world_load(argv[1], &w);  // host only: every rank reads the file and commits the same tree
  if (rc) return rc;
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev == 0) { fprintf(stderr, "crt_rccl_host: no HIP device\n"); return 3; }
  HIP_OK(hipSetDevice(env_int("LOCAL_RANK", rank) % n_dev));
  hipStream_t stream;
  HIP_OK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));

  const bool loopback = env_int("CRT_RCCL_LOOPBACK", 0) != 0;
  ncclUniqueId id;
  ncclComm_t comm = nullptr;
  if (!loopback) {
    if (const int e = exchange_id(&id, rank, world)) return e;
    NCCL_OK(ncclCommInitRank(&comm, world, id, rank));
  }

  const CrtRenderSettings settings = w.settings;
  auto render_rank = [&](int rk) -> CrtRenderer * {
    CrtRenderer *rr = crt_renderer_new(w.scene, w.materials, w.n_geoms, w.lights, w.n_lights, &w.camera, &settings, (uint32_t)rk, (uint32_t)world);
    if (!rr) { fail_lib("crt_renderer_new"); return nullptr; }
    for (uint32_t s = 0; s < w.spp; s += w.batch)
      if (crt_render_samples(rr, s, w.spp - s < w.batch ? w.spp - s : w.batch, stream) != CRT_OK) { fail_lib("crt_render_samples"); crt_renderer_free(rr); return nullptr; }
    return rr;
  };
  CrtRenderer *r = render_rank(loopback ? 0 : rank);
  if (!r) return 3;

  // the shard, zero-padded to the common length; one all-gather; one scatter into the frame
  const size_t own = crt_renderer_pixel_count(r), padded = crt_shard_padded_count(settings.width, settings.height, (uint32_t)world);
  const size_t n_pix = (size_t)settings.width * settings.height;
  if (own > padded) { fprintf(stderr, "crt_rccl_host: shard of %zu pixels, padded length %zu\n", own, padded); return 3; }
  CrtGatherPlan *plan = crt_gather_plan_new(settings.width, settings.height, (uint32_t)world);
  if (!plan || crt_gather_plan_padded_count(plan) != padded) return fail_lib("crt_gather_plan_new");
  float *d_send = nullptr, *d_recv = nullptr, *d_frame = nullptr;
  unsigned long long *d_stats = nullptr;
  HIP_OK(hipMalloc(&d_send, padded * 12));
  HIP_OK(hipMalloc(&d_recv, (size_t)world * padded * 12));
  HIP_OK(hipMalloc(&d_frame, n_pix * 12));
  HIP_OK(hipMalloc(&d_stats, 64));
  HIP_OK(hipMemsetAsync(d_send, 0, padded * 12, stream));
  HIP_OK(hipMemsetAsync(d_frame, 0, n_pix * 12, stream));
  CrtRayStats mine;
  if (!loopback) {
    if (crt_film_resolve(r, d_send, stream) != CRT_OK) return fail_lib("crt_film_resolve");
    NCCL_OK(ncclAllGather(d_send, d_recv, padded * 3, ncclFloat, comm, stream));
    if (crt_render_stats(r, &mine) != CRT_OK) return fail_lib("crt_render_stats");  // drains the batches' stream
    HIP_OK(hipMemcpyAsync(d_stats, &mine, 64, hipMemcpyHostToDevice, stream));
    NCCL_OK(ncclAllReduce(d_stats, d_stats, 8, ncclUint64, ncclSum, comm, stream));
  } else {  // every rank's shard where the all-gather would have put it; the counters' sum on the host
    HIP_OK(hipMemsetAsync(d_recv, 0, (size_t)world * padded * 12, stream));
    unsigned long long sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int rk = 0; rk < world; rk++) {
      CrtRenderer *rr = rk == 0 ? r : render_rank(rk);
      if (!rr) return 3;
      if (crt_renderer_pixel_count(rr) > padded) return 3;
      if (crt_film_resolve(rr, d_recv + (size_t)rk * padded * 3, stream) != CRT_OK) return fail_lib("crt_film_resolve");
      if (crt_render_stats(rr, &mine) != CRT_OK) return fail_lib("crt_render_stats");
      const unsigned long long *m = reinterpret_cast<const unsigned long long *>(&mine);
      for (int k = 0; k < 8; k++) sum[k] += m[k];
      HIP_OK(hipStreamSynchronize(stream));
      if (rk != 0) crt_renderer_free(rr);
    }
    HIP_OK(hipMemcpyAsync(d_stats, sum, 64, hipMemcpyHostToDevice, stream));
    HIP_OK(hipStreamSynchronize(stream));  // `sum` leaves scope
  }
  if (crt_gather_plan_assemble(plan, d_recv, d_frame, stream) != CRT_OK) return fail_lib("crt_gather_plan_assemble");

  std::vector<float> frame(n_pix * 3);
  CrtRayStats total;
  HIP_OK(hipMemcpyAsync(frame.data(), d_frame, n_pix * 12, hipMemcpyDeviceToHost, stream));
  HIP_OK(hipMemcpyAsync(&total, d_stats, 64, hipMemcpyDeviceToHost, stream));
  HIP_OK(hipStreamSynchronize(stream));
  if (rank == 0) {
    FILE *o = fopen(argv[2], "wb");
    if (!o || fwrite(frame.data(), 12, n_pix, o) != n_pix || fwrite(&total, sizeof total, 1, o) != 1) { perror(argv[2]); return 2; }
    fclose(o);
    int ver = 0;
    ncclGetVersion(&ver);
    printf("crt_rccl_host: world %d%s, RCCL %d: %ux%u, %u spp; all-gather of %zu x 3 floats per rank; %llu closest-hit + %llu shadow rays in all\n",
           world, loopback ? " (loopback: one process plays every rank, no communicator)" : "", ver, settings.width, settings.height, w.spp,
           padded, (unsigned long long)total.closest_hit,
           (unsigned long long)total.shadow_rays);
  }
  if (comm) ncclCommDestroy(comm);
  crt_gather_plan_free(plan);
  crt_renderer_free(r);
  (void)hipFree(d_send); (void)hipFree(d_recv); (void)hipFree(d_frame); (void)hipFree(d_stats);
  (void)hipStreamDestroy(stream);
  world_free(&w);
  return 0;
}
