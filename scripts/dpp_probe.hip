#include <hip/hip_runtime.h>
__device__ __forceinline__ double shr1(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(lo, lo, 0x138, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(hi, hi, 0x138, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double shl1(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(lo, lo, 0x130, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(hi, hi, 0x130, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__global__ void k(double* a, double* b, double* c) {
  int i = threadIdx.x;
  double v = a[i];
  b[i] = shr1(v);   // expect b[i] = a[i-1]
  c[i] = shl1(v);   // expect c[i] = a[i+1]
}
int main(){ double *a,*b,*c; hipMalloc(&a,512);hipMalloc(&b,512);hipMalloc(&c,512);
 double h[64]; for(int i=0;i<64;i++)h[i]=i; hipMemcpy(a,h,512,hipMemcpyHostToDevice);
 k<<<1,64>>>(a,b,c); double hb[64],hc[64]; hipMemcpy(hb,b,512,hipMemcpyDeviceToHost); hipMemcpy(hc,c,512,hipMemcpyDeviceToHost);
 for(int i=0;i<64;i++) printf("%d: shr=%g shl=%g\n", i, hb[i], hc[i]); }
