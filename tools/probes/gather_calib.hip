// gather_calib.hip -- calibration of rocprofv3's FETCH_SIZE and of the vector-memory address path on gfx950 for the access shapes
// of the traversal kernels: per-lane gathers of 16 / 32 / 64-byte records at random indices (the per-lane
// walk's node fetches: 1, 2 or 4 global_load_dwordx4 per lane and record) and wave-uniform 64-byte scalar loads (the packet
// walks' s_load_dwordx16), against a wide coalesced stream (the shape MI355X_MICROARCH.md's x2 correction is stated for).
//
//   hipcc --offload-arch=gfx950 -O3 -o gather_calib gather_calib.hip
//   ./gather_calib                              -> timing table (records/s, lane-loads per clock per CU)
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -- ./gather_calib pmc
//                                               -> FETCH_SIZE per dispatch; tools/probes/gather_calib_summary.py divides by the
//                                                  byte counts this program prints ("bytes" lines)
// Every kernel reads a known number of records; `sink` keeps the loads alive.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t pcg(uint32_t& s) {
  s = s * 747796405u + 2891336453u;
  const uint32_t w = ((s >> ((s >> 28u) + 4u)) ^ s) * 277803737u;
  return (w >> 22u) ^ w;
}

// wide coalesced stream: every lane 16 bytes, consecutive lanes consecutive addresses
__global__ void k_stream16(const float4* __restrict__ a, uint64_t n, float* sink) {
  float acc = 0.0f;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const float4 v = a[i];
    acc += v.x + v.y + v.z + v.w;
  }
  if (acc == 12345.678f) *sink = acc;
}

// per-lane gather of RECORD-byte records (RECORD / 16 dwordx4 loads per lane and record) at random record indices
template <int F4>
__global__ void k_gather(const float4* __restrict__ a, uint32_t records, uint32_t perLane, float* sink) {
  uint32_t s = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 12345u;
  float acc = 0.0f;
  for (uint32_t k = 0; k < perLane; k++) {
    const uint32_t r = pcg(s) % records;
    const float4* p = a + (size_t)r * F4;
#pragma unroll
    for (int j = 0; j < F4; j++) {
      const float4 v = p[j];
      acc += v.x + v.w;
    }
  }
  if (acc == 12345.678f) *sink = acc;
}

// the same as a DEPENDENT chain (the next index comes out of the record just read: a pointer chase, as a tree walk is)
template <int F4>
__global__ void k_chase(const float4* __restrict__ a, uint32_t records, uint32_t perLane, float* sink) {
  uint32_t r = ((blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 12345u) % records;
  float acc = 0.0f;
  for (uint32_t k = 0; k < perLane; k++) {
    const float4* p = a + (size_t)r * F4;
    float4 v = p[0];
#pragma unroll
    for (int j = 1; j < F4; j++) {
      const float4 w = p[j];
      acc += w.x + w.w;
    }
    acc += v.y;
    r = __float_as_uint(v.x) % records;   // (the table holds random integers in .x)
  }
  if (acc == 12345.678f) *sink = acc;
}

// wave-uniform 64-byte scalar loads at random record indices (one s_load_dwordx16 per wave and record)
typedef float F16v __attribute__((ext_vector_type(16)));
typedef const __attribute__((address_space(4))) F16v* ConstF16;
__global__ void k_sload64(const float4* __restrict__ a, uint32_t records, uint32_t perWave, float* sink) {
  uint32_t s = (uint32_t)__builtin_amdgcn_readfirstlane((int)((blockIdx.x * blockDim.x + threadIdx.x) / 64u)) * 2654435761u + 777u;
  float acc = 0.0f;
  const ConstF16 base = (ConstF16)(unsigned long long)a;
  for (uint32_t k = 0; k < perWave; k++) {
    const uint32_t r = (uint32_t)__builtin_amdgcn_readfirstlane((int)(pcg(s) % records));
    const F16v v = base[r];
    acc += v.s0 + v.sf;
  }
  if (acc == 12345.678f) *sink = acc;
}

struct Timer {
  hipEvent_t a, b;
  Timer() { CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b)); }
  void start() { CHECK(hipEventRecord(a, 0)); }
  float stop() { CHECK(hipEventRecord(b, 0)); CHECK(hipEventSynchronize(b)); float ms; CHECK(hipEventElapsedTime(&ms, a, b)); return ms; }
};

int main(int argc, char** argv) {
  const bool pmc = argc > 1 && !strcmp(argv[1], "pmc");   // one launch of each kernel, no repeats (counter passes)
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  const double ghz = prop.clockRate * 1e-6;
  printf("device %s, %d CUs, %.2f GHz nominal\n", prop.gcnArchName, cus, ghz);
  float* sink;
  CHECK(hipMalloc(&sink, 4));
  Timer t;
  const uint64_t sizes[] = {16ull << 10, 2ull << 20, 24ull << 20, 128ull << 20, 1ull << 30, 4ull << 30};
  const char* names[] = {"16 KiB (L1)", "2 MiB (L2)", "24 MiB (8 x L2)", "128 MiB (Infinity Cache)", "1 GiB (HBM)", "4 GiB (HBM)"};
  const uint64_t maxBytes = sizes[5];
  float4* table;
  CHECK(hipMalloc(&table, maxBytes));
  {   // random integers in every .x (the chase kernels' next index), anything elsewhere
    std::vector<uint32_t> h((size_t)(64u << 20) / 4);
    uint32_t s = 99u;
    for (auto& w : h) { s = s * 1664525u + 1013904223u; w = s >> 4; }
    for (uint64_t off = 0; off < maxBytes; off += (64ull << 20)) CHECK(hipMemcpy((char*)table + off, h.data(), 64u << 20, hipMemcpyHostToDevice));
  }
  const dim3 block(64);
  const dim3 grid((uint32_t)cus * 32u);   // every wave slot once (8 waves per SIMD)
  const uint64_t lanes = (uint64_t)grid.x * 64;

  // ---- wide coalesced stream (the guide's calibrated shape)
  for (int si = 3; si < 6; si++) {
    const uint64_t n = sizes[si] / 16;
    if (!pmc) { k_stream16<<<grid, dim3(256)>>>(table, n, sink); CHECK(hipDeviceSynchronize()); }
    t.start();
    k_stream16<<<grid, dim3(256)>>>(table, n, sink);
    const float ms = t.stop();
    printf("bytes k_stream16 table=%s bytes=%llu ms=%.3f GB/s=%.0f\n", names[si], (unsigned long long)sizes[si], ms, sizes[si] / ms * 1e-6);
    if (pmc) break;
  }

  // ---- per-lane gathers
  const uint32_t perLane = pmc ? 64u : 256u;
  auto run_gather = [&](int f4, bool chase, int si) {
    const uint32_t records = (uint32_t)(sizes[si] / (16u * f4));
    auto launch = [&]() {
      if (!chase) {
        if (f4 == 1) k_gather<1><<<grid, block>>>(table, records, perLane, sink);
        else if (f4 == 2) k_gather<2><<<grid, block>>>(table, records, perLane, sink);
        else k_gather<4><<<grid, block>>>(table, records, perLane, sink);
      } else {
        if (f4 == 1) k_chase<1><<<grid, block>>>(table, records, perLane, sink);
        else if (f4 == 2) k_chase<2><<<grid, block>>>(table, records, perLane, sink);
        else k_chase<4><<<grid, block>>>(table, records, perLane, sink);
      }
    };
    if (!pmc) { launch(); CHECK(hipDeviceSynchronize()); }
    t.start();
    launch();
    const float ms = t.stop();
    const double recs = (double)lanes * perLane;
    const double laneLoads = recs * f4;
    printf("bytes %s<%d> table=%s records=%.0f record_bytes=%d bytes=%.0f ms=%.3f Grecords/s=%.2f lane-loads/clk/CU=%.3f (at %.2f GHz) GB/s(records)=%.0f\n",
           chase ? "k_chase" : "k_gather", f4, names[si], recs, 16 * f4, recs * 16 * f4, ms, recs / ms * 1e-6, laneLoads / (ms * 1e-3) / (ghz * 1e9) / cus, ghz,
           recs * 16 * f4 / ms * 1e-6);
  };
  for (int si = 0; si < 6; si++) {
    if (pmc && si != 3 && si != 4) continue;   // counter passes: the 128 MiB and the 1 GiB tables
    for (int f4 : {1, 2, 4}) {
      run_gather(f4, false, si);
      run_gather(f4, true, si);
    }
  }

  // ---- wave-uniform scalar loads
  const uint32_t perWave = pmc ? 2048u : 8192u;
  for (int si = 1; si < 5; si++) {
    if (pmc && si != 3 && si != 4) continue;
    const uint32_t records = (uint32_t)(sizes[si] / 64u);
    if (!pmc) { k_sload64<<<grid, block>>>(table, records, perWave, sink); CHECK(hipDeviceSynchronize()); }
    t.start();
    k_sload64<<<grid, block>>>(table, records, perWave, sink);
    const float ms = t.stop();
    const double recs = (double)grid.x * perWave;
    printf("bytes k_sload64 table=%s records=%.0f record_bytes=64 bytes=%.0f ms=%.3f Grecords/s=%.3f GB/s(records)=%.0f\n", names[si], recs, recs * 64, ms,
           recs / ms * 1e-6, recs * 64 / ms * 1e-6);
  }
  return 0;
}
