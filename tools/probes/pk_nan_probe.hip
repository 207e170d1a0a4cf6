// Probe: are gfx950's packed-f32 VALU ops IEEE for the inf*0 / NaN-propagation cases the BVH box test
// depends on (acc.cl:113-130 relies on 0*inf = NaN and on NaN compares being false)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef float f2 __attribute__((ext_vector_type(2)));
__global__ void probe(const float* in, float* out) {
  f2 a = {in[0], in[1]}, b = {in[2], in[3]}, r;
  asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  out[0] = r.x; out[1] = r.y;
  float s0, s1;
  asm volatile("v_mul_f32 %0, %1, %2" : "=v"(s0) : "v"(in[0]), "v"(in[2]));
  asm volatile("v_mul_f32 %0, %1, %2" : "=v"(s1) : "v"(in[1]), "v"(in[3]));
  out[2] = s0; out[3] = s1;
  f2 c = {in[4], in[5]}, d = {in[6], in[7]}, q;
  asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(q) : "v"(c), "v"(d));
  out[4] = q.x; out[5] = q.y;
  out[6] = in[4] - in[6]; out[7] = in[5] - in[7];
}
int main() {
  float h[8] = {0.0f, INFINITY, INFINITY, 0.0f, 0.0f, 2.5f, 0.0f, 2.5f}, o[8], *di, *dout;
  hipMalloc(&di, sizeof(h)); hipMalloc(&dout, sizeof(o));
  hipMemcpy(di, h, sizeof(h), hipMemcpyHostToDevice);
  probe<<<1, 1>>>(di, dout);
  hipMemcpy(o, dout, sizeof(o), hipMemcpyDeviceToHost);
  printf("v_pk_mul_f32(0*inf, inf*0) = %a %a   v_mul_f32 = %a %a\n", o[0], o[1], o[2], o[3]);
  printf("v_pk_add_f32(0-0, 2.5-2.5) = %a %a   scalar = %a %a\n", o[4], o[5], o[6], o[7]);
  return 0;
}
