// Layout probe for v_mfma_f64_4x4x4_4b_f64: one-hot A (lane la) x one-hot B (lane lb) -> which D lanes light up.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(double* D){ // grid 64x64, block 64
  int la=blockIdx.x, lb=blockIdx.y, l=threadIdx.x;
  double a=(l==la)?1.0:0.0, b=(l==lb)?1.0:0.0, c=0.0;
  c=__builtin_amdgcn_mfma_f64_4x4x4f64(a,b,c,0,0,0);
  D[(la*64+lb)*64+l]=c;
}
int main(){
  double* d; hipMalloc(&d,8*64*64*64);
  k<<<dim3(64,64),64>>>(d);
  std::vector<double> h(64*64*64); hipMemcpy(h.data(),d,8*64*64*64,hipMemcpyDeviceToHost);
  // For each la: list lb's that produce output and the output lane
  for(int la=0; la<64; ++la){
    printf("la=%2d:",la);
    for(int lb=0; lb<64; ++lb){
      for(int l=0;l<64;l++) if(h[(la*64+lb)*64+l]!=0.0) printf(" (lb=%d->d=%d)",lb,l);
    }
    printf("\n");
  }
  return 0;
}
