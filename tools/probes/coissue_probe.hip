// Probe: do f64 MFMA (4x4x4) and VALU (f64 FMA / 32-bit DPP mov / ds_bpermute) from two different waves of
// one SIMD overlap on gfx950?  512-thread workgroups, one per CU: waves w and w+4 share a SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
template<int MODE_A, int MODE_B>   // mode: 0 idle, 1 mfma, 2 fma64, 3 dpp mov, 4 bpermute, 5 mixed mfma+dpp same wave
__global__ void __launch_bounds__(512) k(double* out, int iters){
  const int wave = threadIdx.x >> 6;
  const int mode = (wave < 4) ? MODE_A : MODE_B;
  double a = 1.0 + threadIdx.x*1e-3, b = 1.0 - threadIdx.x*1e-3;
  double acc[12]; for(int i=0;i<12;i++) acc[i]=i;
  int iv[8]; for(int i=0;i<8;i++) iv[i]=threadIdx.x+i;
  const int addr = ((threadIdx.x & ~15) | ((threadIdx.x + 4) & 15)) * 4;
  if (mode == 1) {
    for(int it=0; it<iters; ++it){
#pragma unroll
      for(int i=0;i<12;i++) acc[i]=__builtin_amdgcn_mfma_f64_4x4x4f64(a,b,acc[i],0,0,0);
    }
  } else if (mode == 2) {
    for(int it=0; it<iters; ++it){
#pragma unroll
      for(int i=0;i<12;i++) acc[i]=__builtin_fma(a,acc[i],b);
    }
  } else if (mode == 3) {
    for(int it=0; it<iters; ++it){
#pragma unroll
      for(int i=0;i<8;i++) iv[i]=__builtin_amdgcn_mov_dpp(iv[i]+1, 0x120+12, 0xf, 0xf, true);
    }
  } else if (mode == 4) {
    for(int it=0; it<iters; ++it){
#pragma unroll
      for(int i=0;i<8;i++) iv[i]=__builtin_amdgcn_ds_bpermute(addr, iv[i]);
    }
  } else if (mode == 5) {
    for(int it=0; it<iters; ++it){
#pragma unroll
      for(int i=0;i<12;i++){ acc[i]=__builtin_amdgcn_mfma_f64_4x4x4f64(a,b,acc[i],0,0,0);
        if (i<8) iv[i]=__builtin_amdgcn_mov_dpp(iv[i], 0x120+12, 0xf, 0xf, true); }
    }
  } else if (mode == 6) {
    for(int it=0; it<iters; ++it){
#pragma unroll
      for(int i=0;i<12;i++){ acc[i]=__builtin_amdgcn_mfma_f64_4x4x4f64(a,b,acc[i],0,0,0);
        if (i<6) { double t=__builtin_fma(a,(double)iv[i],b); iv[i]=__double2loint(t);} }
    }
  }
  double s=0; for(int i=0;i<12;i++) s+=acc[i]; for(int i=0;i<8;i++) s+=iv[i];
  out[blockIdx.x*512+threadIdx.x]=s;
}
template<int A,int B> float run(double* out,int iters){
  hipEvent_t e0,e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<A,B><<<256,512>>>(out,iters); hipDeviceSynchronize();
  hipEventRecord(e0); k<A,B><<<256,512>>>(out,iters); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms,e0,e1); return ms;
}
int main(){
  double* out; hipMalloc(&out, 8*512*256);
  const int it=40000;
  printf("per-instruction cycles @2.4GHz assume 12 (mfma,fma) or 8 (dpp,bperm) instr per iteration\n");
  float m=run<1,0>(out,it); printf("mfma alone            : %.3f ms  (%.1f cyc/mfma)\n",m,m*1e-3*2.4e9/(it*12.0));
  float v=run<0,2>(out,it); printf("fma64 alone           : %.3f ms  (%.1f cyc/fma)\n",v,v*1e-3*2.4e9/(it*12.0));
  float d=run<0,3>(out,it); printf("dpp(add+mov) alone    : %.3f ms  (%.1f cyc/pair)\n",d,d*1e-3*2.4e9/(it*8.0));
  float p=run<0,4>(out,it); printf("bpermute alone        : %.3f ms  (%.1f cyc/op)\n",p,p*1e-3*2.4e9/(it*8.0));
  float mv=run<1,2>(out,it); printf("mfma || fma64         : %.3f ms  (sum %.3f, max %.3f)\n",mv,m+v,m>v?m:v);
  float md=run<1,3>(out,it); printf("mfma || dpp           : %.3f ms  (sum %.3f, max %.3f)\n",md,m+d,m>d?m:d);
  float mp=run<1,4>(out,it); printf("mfma || bpermute      : %.3f ms  (sum %.3f, max %.3f)\n",mp,m+p,m>p?m:p);
  float mm=run<1,1>(out,it); printf("mfma || mfma          : %.3f ms  (2x alone %.3f)\n",mm,2*m);
  float x5=run<5,0>(out,it); printf("same wave mfma+8dpp/12: %.3f ms  (mfma alone %.3f)\n",x5,m);
  float x55=run<5,5>(out,it); printf("2 waves mfma+8dpp/12  : %.3f ms  (2x mfma alone %.3f)\n",x55,2*m);
  float x6=run<6,0>(out,it); printf("same wave mfma+6fma/12: %.3f ms  (mfma alone %.3f)\n",x6,m);
  float x66=run<6,6>(out,it); printf("2 waves mfma+6fma/12  : %.3f ms  (2x mfma alone %.3f)\n",x66,2*m);
  return 0;
}
