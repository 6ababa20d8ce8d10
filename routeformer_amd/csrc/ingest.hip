// Video ingest on the GPU: the dataset-side work in front of the conv trunk.
//   * rf_resize_area    -- `cv2.resize(frame, target, interpolation=cv2.INTER_AREA)` of io/dataset.py:1476-1497 for
//                          down-scaling factors < 1: every output pixel is the coverage-weighted mean of the source
//                          pixels its footprint touches (box filter; integer factors = plain s x s block means).
//   * rf_frame_hash     -- 64-bit content hash per frame: the key of the backbone-feature cache that replaces
//                          `@torchcache(persistent=True)` (models/video_backbone/__init__.py:14-32) -- the frozen
//                          trunk's tokens of a frame are looked up by what the frame CONTAINS, never by an address.
//   * rf_cache_lookup / rf_cache_insert -- an open-addressing table key -> slot in device memory (the token slots
//                          themselves are a plain [capacity][65][240] tensor in HBM: at 62 KB per frame 288 GB hold
//                          millions of frames, i.e. the whole dataset stays resident after the first epoch).
// All byte / index work: HBM-bound, coalesced, no matrix cores.
#include "common.h"

namespace {

// ---- area resize (uint8, planar frames [N][H][W] -> [N][h][w]) -------------------------------------------------------
// Source footprint of output pixel (y, x): rows [y * sy, (y + 1) * sy), columns [x * sx, (x + 1) * sx) with
// sy = H / h, sx = W / w (real numbers); a source pixel contributes with the fraction of it that lies inside.
__global__ __launch_bounds__(256) void resize_area_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, long N,
                                                          int H, int W, int h, int w) {
  const double sy = (double)H / h, sx = (double)W / w;
  const float inv_area = (float)(1.0 / (sy * sx));
  const long total = N * h * w;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int x = (int)(i % w), y = (int)((i / w) % h);
    const long n = i / ((long)w * h);
    if (H == 2 * h && W == 2 * w) {
      // exact factor 1/2: OpenCV takes its 2 x 2 fast path here (resize.cpp, ResizeAreaFastVec: (a + b + c + d + 2) >> 2),
      // which rounds a half UP, where the general path's saturate_cast rounds it to even -- a quarter of the pixels of a
      // natural frame sit on such a half
      const uint8_t* p0 = src + n * (long)H * W + (long)(2 * y) * W + 2 * x;
      dst[i] = (uint8_t)(((int)p0[0] + (int)p0[1] + (int)p0[W] + (int)p0[W + 1] + 2) >> 2);
      continue;
    }
    const double fy0 = y * sy, fy1 = (y + 1) * sy, fx0 = x * sx, fx1 = (x + 1) * sx;
    const int y0 = (int)fy0, y1 = min(H, (int)ceil(fy1 - 1e-9)), x0 = (int)fx0, x1 = min(W, (int)ceil(fx1 - 1e-9));
    const uint8_t* img = src + n * (long)H * W;
    float acc = 0.f;
    for (int yy = y0; yy < y1; ++yy) {
      const float wy = (float)(fmin((double)yy + 1, fy1) - fmax((double)yy, fy0));
      float row = 0.f;
      for (int xx = x0; xx < x1; ++xx) {
        const float wx = (float)(fmin((double)xx + 1, fx1) - fmax((double)xx, fx0));
        row = fmaf(wx, (float)img[(long)yy * W + xx], row);
      }
      acc = fmaf(wy, row, acc);
    }
    dst[i] = (uint8_t)fminf(255.f, fmaxf(0.f, rintf(acc * inv_area)));  // round half to even, saturate (cv::saturate_cast)
  }
}

// ---- content hash ----------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long mix64(unsigned long long x) {  // splitmix64 finaliser
  x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
  x ^= x >> 27; x *= 0x94D049BB133111EBull;
  x ^= x >> 31;
  return x;
}

// Every 8-byte word is mixed with its position (so permutations of the content change the key) and the terms are ADDED
// (wrap-around: order-independent), so a frame's words can be split over any number of workgroups: gridDim.y slices per
// frame, each adds its partial sum into keys[frame] (zeroed by the caller's launch sequence), frame_hash_final_kernel
// mixes the total.  (One workgroup per frame was 280 us for 64 frames of 448 x 448: 64 workgroups walking 1.2 MB each.)
// The key depends on the bytes only -- not on the slicing.
// Frame f of clip [B][F][bytes_per_frame] (contiguous); tail bytes (bytes % 8) are folded in as one padded word.
__global__ __launch_bounds__(256) void frame_hash_kernel(const unsigned char* __restrict__ data, long bytes_per_frame,
                                                         const int64_t* __restrict__ frame_ids,
                                                         unsigned long long* __restrict__ keys, unsigned long long seed) {
  __shared__ unsigned long long part[256];
  const unsigned char* p = data + (frame_ids ? (long)frame_ids[blockIdx.x] : (long)blockIdx.x) * bytes_per_frame;
  const long words = bytes_per_frame >> 3;
  const long per = (words + gridDim.y - 1) / gridDim.y, w0 = (long)blockIdx.y * per, w1 = min(words, w0 + per);
  unsigned long long h = 0;
  const bool aligned = (reinterpret_cast<uintptr_t>(p) & 7) == 0;
  for (long i = w0 + threadIdx.x; i < w1; i += 256) {
    unsigned long long v;
    if (aligned) v = reinterpret_cast<const unsigned long long*>(p)[i];
    else { v = 0; for (int b = 0; b < 8; ++b) v |= (unsigned long long)p[i * 8 + b] << (8 * b); }
    h += mix64(v ^ mix64((unsigned long long)i + seed));
  }
  if (threadIdx.x == 0 && blockIdx.y == 0 && (bytes_per_frame & 7)) {
    unsigned long long v = 0;
    for (int b = 0; b < (int)(bytes_per_frame & 7); ++b) v |= (unsigned long long)p[words * 8 + b] << (8 * b);
    h += mix64(v ^ mix64((unsigned long long)words + seed));
  }
  part[threadIdx.x] = h;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) part[threadIdx.x] += part[threadIdx.x + s];  // wrap-around addition: order-independent
    __syncthreads();
  }
  if (threadIdx.x == 0) atomicAdd(&keys[blockIdx.x], part[0]);
}

__global__ void frame_hash_final_kernel(unsigned long long* __restrict__ keys, int n, long bytes_per_frame) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const unsigned long long k = mix64(keys[i] ^ ((unsigned long long)bytes_per_frame * 0x9E3779B97F4A7C15ull));
  keys[i] = k == 0 ? 1 : k;  // 0 marks an empty table entry
}

// ---- key -> slot table (open addressing, linear probing; capacity a power of two; key 0 = empty) ----------------------
__global__ void cache_lookup_kernel(const unsigned long long* __restrict__ keys, int n, const unsigned long long* __restrict__ table_keys,
                                    const int32_t* __restrict__ table_slots, int cap_mask, int32_t* __restrict__ slots,
                                    int32_t* __restrict__ misses) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const unsigned long long k = keys[i];
  int pos = (int)(k & (unsigned long long)cap_mask), found = -1;
  for (int probe = 0; probe <= cap_mask; ++probe) {
    const unsigned long long t = table_keys[pos];
    if (t == k) { found = table_slots[pos]; break; }
    if (t == 0) break;
    pos = (pos + 1) & cap_mask;
  }
  slots[i] = found;
  if (found < 0) atomicAdd(misses, 1);
}

// Insert the keys whose slots[i] < 0: claim a table entry (atomicCAS on the key), take the next free token slot from
// `next_slot` (atomic counter) and publish it; slots[i] = the new slot (its tokens are written by the caller).  A key
// that another thread of the same launch claims first (identical frames in one batch) is left at -1 here and resolved
// by the rf_cache_lookup the caller issues afterwards -- no thread ever waits for another one.  A full cache (no slot
// or no table entry left) also leaves -1: such frames simply stay uncached.
__global__ void cache_insert_kernel(const unsigned long long* __restrict__ keys, int n, unsigned long long* __restrict__ table_keys,
                                    int32_t* __restrict__ table_slots, int cap_mask, int32_t* __restrict__ next_slot, int n_slots,
                                    int32_t* __restrict__ slots) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || slots[i] >= 0) return;
  if (*reinterpret_cast<volatile int32_t*>(next_slot) >= n_slots) return;  // full: do not burn table entries
  const unsigned long long k = keys[i];
  int pos = (int)(k & (unsigned long long)cap_mask);
  for (int probe = 0; probe <= cap_mask; ++probe) {
    const unsigned long long prev = atomicCAS(&table_keys[pos], 0ull, k);
    if (prev == 0ull) {  // claimed an empty entry
      const int s = atomicAdd(next_slot, 1);
      if (s < n_slots) {
        table_slots[pos] = s;
        slots[i] = s;
      } else {
        table_slots[pos] = -1;  // (entry burnt: the key maps to "no slot")
      }
      return;
    }
    if (prev == k) return;  // owned by another thread / an earlier call
    pos = (pos + 1) & cap_mask;
  }
}

// ---- frame sub-sampling into staging buffers: dst[b][f] = src[b][idx[f]] for up to 8 clips in one launch ---------------
struct GatherTable {
  int count, pad;
  RfGatherEntry e[RF_GATHER_MAX];
  long first_block[RF_GATHER_MAX + 1];
};
constexpr int GF_CHUNK = 16384;  // bytes per workgroup (256 threads x 4 x 16 B)

__global__ __launch_bounds__(256) void gather_frames_kernel(const GatherTable t) {
  int ei = 0;
  while (ei + 1 < t.count && (long)blockIdx.x >= t.first_block[ei + 1]) ++ei;
  const RfGatherEntry& e = t.e[ei];
  const long local = (long)blockIdx.x - t.first_block[ei];
  const long chunks = (e.frame_bytes + GF_CHUNK - 1) / GF_CHUNK;
  const long frame = local / chunks, c = local - frame * chunks;
  const int b = (int)(frame / e.F), f = (int)(frame - (long)b * e.F);
  const long src_frame = (long)b * e.T + e.idx[f];
  const unsigned char* src = static_cast<const unsigned char*>(e.src) + src_frame * e.frame_bytes + c * GF_CHUNK;
  unsigned char* dst = static_cast<unsigned char*>(e.dst) + frame * e.frame_bytes + c * GF_CHUNK;
  const long n = min((long)GF_CHUNK, e.frame_bytes - c * GF_CHUNK);
  if (((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0) {
    const long n16 = n >> 4;
    for (long i = threadIdx.x; i < n16; i += 256)
      reinterpret_cast<float4*>(dst)[i] = reinterpret_cast<const float4*>(src)[i];
    for (long i = (n16 << 4) + threadIdx.x; i < n; i += 256) dst[i] = src[i];
  } else {
    for (long i = threadIdx.x; i < n; i += 256) dst[i] = src[i];
  }
}

inline bool pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

}  // namespace

extern "C" int rf_gather_frames(const RfGatherEntry* entries, int count, void* stream) {
  RF_REQUIRE(entries && count >= 1 && count <= RF_GATHER_MAX);
  GatherTable t{};
  t.count = count;
  long blocks = 0;
  for (int i = 0; i < count; ++i) {
    const RfGatherEntry& e = entries[i];
    RF_REQUIRE(e.src && e.dst && e.idx && e.B > 0 && e.T > 0 && e.F > 0 && e.frame_bytes > 0);
    t.e[i] = e;
    t.first_block[i] = blocks;
    blocks += (long)e.B * e.F * ((e.frame_bytes + GF_CHUNK - 1) / GF_CHUNK);
  }
  t.first_block[count] = blocks;
  RF_REQUIRE(blocks < (1L << 31));
  RF_LAUNCH(gather_frames_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), t);
  RF_CHECK_LAUNCH();
  return RF_OK;
}

extern "C" int rf_resize_area(const uint8_t* src, uint8_t* dst, int64_t n_planes, int H, int W, int h, int w, void* stream) {
  RF_REQUIRE(src && dst && n_planes > 0 && H > 0 && W > 0 && h > 0 && w > 0 && h <= H && w <= W);
  const long total = n_planes * (long)h * w;
  const int blocks = (int)((total + 255) / 256 > 65535 ? 65535 : (total + 255) / 256);
  RF_LAUNCH(resize_area_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), src, dst, (long)n_planes, H, W, h, w);
  RF_CHECK_LAUNCH();
  return RF_OK;
}

extern "C" int rf_frame_hash(const void* frames, const int64_t* frame_ids, int64_t n_frames, int64_t bytes_per_frame,
                             uint64_t* keys, int64_t seed, void* stream) {
  RF_REQUIRE(frames && keys && n_frames > 0 && n_frames < (1 << 30) && bytes_per_frame > 0);
  hipStream_t st = static_cast<hipStream_t>(stream);
  // slices per frame: towards ~2k workgroups, at least 16 KB of frame each
  long slices = (2048 + n_frames - 1) / n_frames;
  const long most = bytes_per_frame / (16 * 1024);
  if (slices > most) slices = most;
  if (slices < 1) slices = 1;
  if (slices > 1024) slices = 1024;
  if (hipMemsetAsync(keys, 0, (size_t)n_frames * sizeof(uint64_t), st) != hipSuccess) {
    rf_g_last_error = "rf_frame_hash: hipMemsetAsync failed";
    return RF_ELAUNCH;
  }
  RF_LAUNCH(frame_hash_kernel, dim3((int)n_frames, (int)slices), dim3(256), 0, st,
            static_cast<const unsigned char*>(frames), (long)bytes_per_frame, frame_ids,
            reinterpret_cast<unsigned long long*>(keys), (unsigned long long)seed);
  RF_CHECK_LAUNCH();
  RF_LAUNCH(frame_hash_final_kernel, dim3((int)((n_frames + 127) / 128)), dim3(128), 0, st,
            reinterpret_cast<unsigned long long*>(keys), (int)n_frames, (long)bytes_per_frame);
  RF_CHECK_LAUNCH();
  return RF_OK;
}

extern "C" int rf_cache_lookup(const uint64_t* keys, int n, const uint64_t* table_keys, const int32_t* table_slots, int capacity,
                               int32_t* slots, int32_t* misses, void* stream) {
  RF_REQUIRE(keys && table_keys && table_slots && slots && misses && n > 0 && pow2(capacity));
  RF_LAUNCH(cache_lookup_kernel, dim3((n + 127) / 128), dim3(128), 0, static_cast<hipStream_t>(stream),
            reinterpret_cast<const unsigned long long*>(keys), n, reinterpret_cast<const unsigned long long*>(table_keys),
            table_slots, capacity - 1, slots, misses);
  RF_CHECK_LAUNCH();
  return RF_OK;
}

extern "C" int rf_cache_insert(const uint64_t* keys, int n, uint64_t* table_keys, int32_t* table_slots, int capacity,
                               int32_t* next_slot, int n_slots, int32_t* slots, void* stream) {
  RF_REQUIRE(keys && table_keys && table_slots && next_slot && slots && n > 0 && pow2(capacity) && n_slots > 0);
  RF_LAUNCH(cache_insert_kernel, dim3((n + 127) / 128), dim3(128), 0, static_cast<hipStream_t>(stream),
            reinterpret_cast<const unsigned long long*>(keys), n, reinterpret_cast<unsigned long long*>(table_keys),
            table_slots, capacity - 1, next_slot, n_slots, slots);
  RF_CHECK_LAUNCH();
  return RF_OK;
}
