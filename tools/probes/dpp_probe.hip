#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ double rot16(double x, int) { return x; }
template<int N> __device__ __forceinline__ double row_ror(double x){
  int lo=__double2loint(x), hi=__double2hiint(x);
  lo=__builtin_amdgcn_update_dpp(0,lo,0x120+N,0xf,0xf,false);
  hi=__builtin_amdgcn_update_dpp(0,hi,0x120+N,0xf,0xf,false);
  return __hiloint2double(hi,lo);
}
__global__ void k(double* o){ int l=threadIdx.x; double v=l; o[l]=row_ror<4>(v); o[64+l]=row_ror<8>(v); o[128+l]=row_ror<12>(v);}
int main(){ double* d; hipMalloc(&d,8*192); k<<<1,64>>>(d); double h[192]; hipMemcpy(h,d,8*192,hipMemcpyDeviceToHost);
 for(int r=0;r<3;r++){ printf("ror%d:",4*(r+1)); for(int l=0;l<20;l++) printf(" %g",h[64*r+l]); printf(" ... l=63:%g\n",h[64*r+63]);} return 0;}
