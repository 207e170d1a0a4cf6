// Probe: do gfx950's packed-f32 VALU ops (v_pk_add_f32 / v_pk_mul_f32) give bit for bit what v_sub_f32 / v_mul_f32
// give, including the cases the BVH box test depends on (acc.cl:113-130 relies on 0*inf = NaN and on NaN compares being
// false) and the denormal range?  Every pair from a table of special values plus random bit patterns is pushed through
// (a - b) * c in both forms; prints the number of mismatching results (want 0).
//   hipcc --offload-arch=gfx950 -O1 -ffp-contract=off tools/probes/pk_nan_probe.hip -o tools/probes/pk_nan_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));

__global__ void probe(const float* a, const float* b, const float* c, uint32_t n, uint32_t* packed, uint32_t* scalar) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (2 * i + 1 >= n) return;
  f2 x = {a[2 * i], a[2 * i + 1]}, y = {b[2 * i], b[2 * i + 1]}, z = {c[2 * i], c[2 * i + 1]}, d, r;
  asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(x), "v"(y));
  asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(r) : "v"(z), "v"(d));
  packed[2 * i] = __float_as_uint(r.x);
  packed[2 * i + 1] = __float_as_uint(r.y);
  float s0, s1, m0, m1;
  asm volatile("v_sub_f32 %0, %1, %2" : "=v"(s0) : "v"(a[2 * i]), "v"(b[2 * i]));
  asm volatile("v_sub_f32 %0, %1, %2" : "=v"(s1) : "v"(a[2 * i + 1]), "v"(b[2 * i + 1]));
  asm volatile("v_mul_f32 %0, %1, %2" : "=v"(m0) : "v"(c[2 * i]), "v"(s0));
  asm volatile("v_mul_f32 %0, %1, %2" : "=v"(m1) : "v"(c[2 * i + 1]), "v"(s1));
  scalar[2 * i] = __float_as_uint(m0);
  scalar[2 * i + 1] = __float_as_uint(m1);
}

int main() {
  const float sp[] = {0.0f, -0.0f, 1.0f, -1.0f, 2.5f, INFINITY, -INFINITY, NAN, 1e-38f, -1e-38f, 1.5e-38f, 1e-40f, -3e-42f, 1e-45f,
                      1.17549435e-38f, 3.4e38f, -3.4e38f, 1e-20f, 1e20f, 0.333333343f};
  const int ns = sizeof(sp) / sizeof(sp[0]);
  std::vector<float> a, b, c;
  for (int i = 0; i < ns; i++)
    for (int j = 0; j < ns; j++)
      for (int k = 0; k < ns; k++) { a.push_back(sp[i]); b.push_back(sp[j]); c.push_back(sp[k]); }
  uint32_t s = 12345u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; uint32_t u = s ^ (s >> 13); float f; memcpy(&f, &u, 4); return f; };
  for (int i = 0; i < (1 << 20); i++) { a.push_back(rnd()); b.push_back(rnd()); c.push_back(rnd()); }
  if (a.size() & 1) { a.push_back(0); b.push_back(0); c.push_back(0); }
  const uint32_t n = (uint32_t)a.size();
  float *da, *db, *dc; uint32_t *dp, *ds;
  hipMalloc(&da, n * 4); hipMalloc(&db, n * 4); hipMalloc(&dc, n * 4); hipMalloc(&dp, n * 4); hipMalloc(&ds, n * 4);
  hipMemcpy(da, a.data(), n * 4, hipMemcpyHostToDevice);
  hipMemcpy(db, b.data(), n * 4, hipMemcpyHostToDevice);
  hipMemcpy(dc, c.data(), n * 4, hipMemcpyHostToDevice);
  probe<<<(n / 2 + 255) / 256, 256>>>(da, db, dc, n, dp, ds);
  std::vector<uint32_t> p(n), q(n);
  hipMemcpy(p.data(), dp, n * 4, hipMemcpyDeviceToHost);
  hipMemcpy(q.data(), ds, n * 4, hipMemcpyDeviceToHost);
  uint32_t bad = 0, nanPayload = 0;
  for (uint32_t i = 0; i < n; i++) {
    if (p[i] == q[i]) continue;
    float fp, fq; memcpy(&fp, &p[i], 4); memcpy(&fq, &q[i], 4);
    if (std::isnan(fp) && std::isnan(fq)) { nanPayload++; continue; }   // both NaN: compares behave the same
    if (bad++ < 10) printf("MISMATCH (%a - %a) * %a: packed %a scalar %a\n", a[i], b[i], c[i], fp, fq);
  }
  printf("pk probe: %u cases, %u mismatches, %u NaN-payload-only differences\n", n, bad, nanPayload);
  return bad != 0;
}
