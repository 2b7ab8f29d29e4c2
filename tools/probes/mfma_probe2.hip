// Probe 2: cycle-accurate f64 MFMA / VALU rates vs waves per SIMD (s_memtime) on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef double d4 __attribute__((ext_vector_type(4)));
template<int NACC>
__global__ void __launch_bounds__(256) k_mfma(double* out, unsigned long long* cyc, int iters, double a0, double b0){
  d4 acc[NACC];
  for(int i=0;i<NACC;i++) acc[i]=(d4){0,0,0,0};
  double a=a0+threadIdx.x*1e-3, b=b0-threadIdx.x*1e-3;
  unsigned long long t0=__builtin_amdgcn_s_memtime();
  unsigned long long r0=__builtin_amdgcn_s_memrealtime();
  for(int it=0; it<iters; ++it){
#pragma unroll
    for(int i=0;i<NACC;i++) acc[i]=__builtin_amdgcn_mfma_f64_16x16x4f64(a,b,acc[i],0,0,0);
  }
  double s=0; for(int i=0;i<NACC;i++) s+=acc[i][0]+acc[i][1]+acc[i][2]+acc[i][3];
  out[blockIdx.x*256+threadIdx.x]=s;
  unsigned long long t1=__builtin_amdgcn_s_memtime();
  unsigned long long r1=__builtin_amdgcn_s_memrealtime();
  if((threadIdx.x&63)==0){ cyc[2*(blockIdx.x*4+(threadIdx.x>>6))]=t1-t0; cyc[2*(blockIdx.x*4+(threadIdx.x>>6))+1]=r1-r0;}
}
template<int NACC>
__global__ void __launch_bounds__(256) k_mfma4(double* out, unsigned long long* cyc, int iters, double a0, double b0){
  double acc[NACC];
  for(int i=0;i<NACC;i++) acc[i]=0;
  double a=a0+threadIdx.x*1e-3, b=b0-threadIdx.x*1e-3;
  unsigned long long t0=__builtin_amdgcn_s_memtime();
  unsigned long long r0=__builtin_amdgcn_s_memrealtime();
  for(int it=0; it<iters; ++it){
#pragma unroll
    for(int i=0;i<NACC;i++) acc[i]=__builtin_amdgcn_mfma_f64_4x4x4f64(a,b,acc[i],0,0,0);
  }
  double s=0; for(int i=0;i<NACC;i++) s+=acc[i];
  out[blockIdx.x*256+threadIdx.x]=s;
  unsigned long long t1=__builtin_amdgcn_s_memtime();
  unsigned long long r1=__builtin_amdgcn_s_memrealtime();
  if((threadIdx.x&63)==0){ cyc[2*(blockIdx.x*4+(threadIdx.x>>6))]=t1-t0; cyc[2*(blockIdx.x*4+(threadIdx.x>>6))+1]=r1-r0;}
}
template<int NACC>
__global__ void __launch_bounds__(256) k_fma(double* out, unsigned long long* cyc, int iters, double a0, double b0){
  double acc[NACC];
  for(int i=0;i<NACC;i++) acc[i]=i;
  double a=a0+threadIdx.x*1e-3, b=b0-threadIdx.x*1e-3;
  unsigned long long t0=__builtin_amdgcn_s_memtime();
  unsigned long long r0=__builtin_amdgcn_s_memrealtime();
  for(int it=0; it<iters; ++it){
#pragma unroll
    for(int i=0;i<NACC;i++) acc[i]=__builtin_fma(a,acc[i],b);
  }
  double s=0; for(int i=0;i<NACC;i++) s+=acc[i];
  out[blockIdx.x*256+threadIdx.x]=s;
  unsigned long long t1=__builtin_amdgcn_s_memtime();
  unsigned long long r1=__builtin_amdgcn_s_memrealtime();
  if((threadIdx.x&63)==0){ cyc[2*(blockIdx.x*4+(threadIdx.x>>6))]=t1-t0; cyc[2*(blockIdx.x*4+(threadIdx.x>>6))+1]=r1-r0;}
}
int main(){
  double* out; hipMalloc(&out, 8*256*8192);
  unsigned long long* cyc; hipMalloc(&cyc, 16*4*8192);
  const int CU=256;
  auto report=[&](const char* name,int wps,int nacc,int iters,double flop_per_instr,float ms){
    int nw=CU*4*wps; std::vector<unsigned long long> h(2*nw);
    hipMemcpy(h.data(),cyc,16*nw,hipMemcpyDeviceToHost);
    std::vector<double> c(nw), r(nw); for(int i=0;i<nw;i++){c[i]=(double)h[2*i]; r[i]=(double)h[2*i+1];}
    std::sort(c.begin(),c.end()); std::sort(r.begin(),r.end());
    double cm=c[nw/2], rm=r[nw/2];
    double clk=cm/rm*100e6; // memrealtime = 100 MHz
    double instr=(double)iters*nacc;
    printf("%-10s wps=%d nacc=%2d: wall %.3f ms | shader cyc/instr/wave %.1f -> per SIMD %.1f | clk %.2f GHz | %.1f TF (wall)\n",
      name,wps,nacc,ms,cm/instr,cm/instr/wps,clk*1e-9, nw*instr*flop_per_instr/(ms*1e-3)*1e-12);
  };
  hipEvent_t e0,e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for(int rep=0; rep<2; ++rep)
  for(int wps : {1,2,4,8}){
    int blocks=CU*wps; float ms; int iters=60000/wps;
#define RUN(K,NAME,NACC,FL) \
    K<NACC><<<blocks,256>>>(out,cyc,iters,1.0,1.0); hipDeviceSynchronize(); \
    hipEventRecord(e0); K<NACC><<<blocks,256>>>(out,cyc,iters,1.0,1.0); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms,e0,e1); \
    if(rep==1) report(NAME,wps,NACC,iters,FL,ms);
    RUN(k_mfma,"mfma16x16x4",4,2048.0)
    RUN(k_mfma,"mfma16x16x4",10,2048.0)
    RUN(k_mfma4,"mfma4x4x4",8,512.0)
    RUN(k_fma,"valu_fma",16,128.0)
  }
  return 0;
}
