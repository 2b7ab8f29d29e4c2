// Probe: do CBSZ/ABID (A-block broadcast) work on v_mfma_f64_4x4x4_4b_f64 on gfx950?
// D[blk] = A[abid-selected] x B[blk]?  One-hot A lane la, B = all ones per lane id encoded.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template<int CBSZ,int ABID>
__global__ void k(double* D){ // A lane l holds value 100*(l) ; B lane l holds 1 if (l==blockIdx.x) else 0
  int l=threadIdx.x; int lb=blockIdx.x;
  double a=(double)(l+1), b=(l==lb)?1.0:0.0, c=0.0;
  c=__builtin_amdgcn_mfma_f64_4x4x4f64(a,b,c,CBSZ,ABID,0);
  D[lb*64+l]=c;
}
template<int CBSZ,int ABID> void run(double* d){
  k<CBSZ,ABID><<<64,64>>>(d);
  std::vector<double> h(64*64); hipMemcpy(h.data(),d,8*64*64,hipMemcpyDeviceToHost);
  printf("cbsz=%d abid=%d\n",CBSZ,ABID);
  // for B one-hot at lane lb=(k,blk,j): D lane (i,blk',j) = A(k, srcblk(blk'), i)  -> print which A lane feeds each output
  for(int lb : {0,5,21,42,63}){
    printf("  lb=%2d (k=%d blk=%d j=%d):",lb,lb>>4,(lb>>2)&3,lb&3);
    for(int l=0;l<64;l++) if(h[lb*64+l]!=0.0) printf(" d%d<-a%d",l,(int)h[lb*64+l]-1);
    printf("\n");
  }
}
int main(){ double* d; hipMalloc(&d,8*64*64);
  run<0,0>(d); run<2,0>(d); run<2,1>(d); run<2,3>(d); run<1,0>(d); run<1,1>(d);
  return 0; }
