// Probe: f64 MFMA 16x16x4 rate + layout, f64 VALU FMA rate on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>
typedef double d4 __attribute__((ext_vector_type(4)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__); return 1;}}while(0)

template<int NACC>
__global__ void __launch_bounds__(64) k_mfma(double* out, int iters, double a0, double b0){
  d4 acc[NACC];
  for(int i=0;i<NACC;i++) acc[i]=(d4){0,0,0,0};
  double a=a0+threadIdx.x*1e-3, b=b0-threadIdx.x*1e-3;
  for(int it=0; it<iters; ++it){
#pragma unroll
    for(int i=0;i<NACC;i++) acc[i]=__builtin_amdgcn_mfma_f64_16x16x4f64(a,b,acc[i],0,0,0);
  }
  double s=0; for(int i=0;i<NACC;i++) s+=acc[i][0]+acc[i][1]+acc[i][2]+acc[i][3];
  out[blockIdx.x*64+threadIdx.x]=s;
}
template<int NACC>
__global__ void __launch_bounds__(64) k_fma(double* out, int iters, double a0, double b0){
  double acc[NACC];
  for(int i=0;i<NACC;i++) acc[i]=i;
  double a=a0+threadIdx.x*1e-3, b=b0-threadIdx.x*1e-3;
  for(int it=0; it<iters; ++it){
#pragma unroll
    for(int i=0;i<NACC;i++) acc[i]=__builtin_fma(a,acc[i],b);
  }
  double s=0; for(int i=0;i<NACC;i++) s+=acc[i];
  out[blockIdx.x*64+threadIdx.x]=s;
}
__global__ void k_layout(const double* A, const double* B, double* C){ // A 16x4 row-major, B 4x16 row-major, C 16x16
  int l=threadIdx.x;
  double a=A[(l&15)*4+(l>>4)];
  double b=B[(l>>4)*16+(l&15)];
  d4 c=(d4){0,0,0,0};
  c=__builtin_amdgcn_mfma_f64_16x16x4f64(a,b,c,0,0,0);
  for(int r=0;r<4;r++) C[((l>>4)+4*r)*16+(l&15)]=c[r];
}
template<class F> float timeit(F f){
  hipEvent_t e0,e1; hipEventCreate(&e0); hipEventCreate(&e1);
  f(); hipDeviceSynchronize();
  hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms,e0,e1); return ms;
}
int main(){
  double* out; CK(hipMalloc(&out, 8*64*4096));
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p,0));
  printf("device %s CUs %d clock %d kHz\n", p.name, p.multiProcessorCount, p.clockRate);
  // layout check
  {
    std::vector<double> A(64),B(64),C(256),R(256,0.0);
    for(int i=0;i<64;i++){A[i]=1+i*0.37; B[i]=2-i*0.11+ (i%7);}
    for(int i=0;i<16;i++)for(int j=0;j<16;j++)for(int k=0;k<4;k++) R[i*16+j]+=A[i*4+k]*B[k*16+j];
    double *dA,*dB,*dC; hipMalloc(&dA,512); hipMalloc(&dB,512); hipMalloc(&dC,2048);
    hipMemcpy(dA,A.data(),512,hipMemcpyHostToDevice); hipMemcpy(dB,B.data(),512,hipMemcpyHostToDevice);
    k_layout<<<1,64>>>(dA,dB,dC); hipMemcpy(C.data(),dC,2048,hipMemcpyDeviceToHost);
    double err=0; for(int i=0;i<256;i++) err=fmax(err,fabs(C[i]-R[i]));
    printf("layout max err %g\n",err);
  }
  const int iters=20000;
  // grid: waves per SIMD = 1 -> 4 waves per CU -> blocks = CUs*4
  for(int wps=1; wps<=2; ++wps){
    int blocks=p.multiProcessorCount*4*wps;
    float ms;
    ms=timeit([&]{k_mfma<1><<<blocks,64>>>(out,iters,1.0,1.0);});
    printf("mfma f64 16x16x4 NACC=1 wps=%d: %.3f ms  -> %.1f cyc/instr@2.4GHz/SIMD, %.2f TF\n",wps,ms, ms*1e-3*2.4e9/(iters*1.0*wps), blocks*(double)iters*1*2048/ms*1e-9);
    ms=timeit([&]{k_mfma<4><<<blocks,64>>>(out,iters,1.0,1.0);});
    printf("mfma f64 16x16x4 NACC=4 wps=%d: %.3f ms  -> %.1f cyc/instr, %.2f TF\n",wps,ms, ms*1e-3*2.4e9/(iters*4.0*wps), blocks*(double)iters*4*2048/ms*1e-9);
    ms=timeit([&]{k_mfma<10><<<blocks,64>>>(out,iters,1.0,1.0);});
    printf("mfma f64 16x16x4 NACC=10 wps=%d: %.3f ms -> %.1f cyc/instr, %.2f TF\n",wps,ms, ms*1e-3*2.4e9/(iters*10.0*wps), blocks*(double)iters*10*2048/ms*1e-9);
    ms=timeit([&]{k_fma<16><<<blocks,64>>>(out,iters,0.999,1.0);});
    printf("valu fma f64 NACC=16 wps=%d: %.3f ms -> %.1f cyc/instr, %.2f TF\n",wps,ms, ms*1e-3*2.4e9/(iters*16.0*wps), blocks*(double)iters*16*128/ms*1e-9);
  }
  return 0;
}
