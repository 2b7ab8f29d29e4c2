#include <hip/hip_runtime.h>
struct PdhDev;
#define S(g) extern "C" hipError_t pdh_launch_g##g(int,int,int,int,int,const PdhDev*,int,size_t,hipStream_t){return hipSuccess;}
S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)
extern "C" hipError_t pdh_launch_rhs(int,int,const PdhDev*,int,const double*,const double*,double*,hipStream_t){return hipSuccess;}
extern "C" hipError_t pdh_launch_eval(int,int,int,const PdhDev*,int,const double*,const int64_t*,const double*,int64_t,double*,double*,hipStream_t){return hipSuccess;}
extern "C" hipError_t pdh_launch_shape(int,int,const PdhDev*,int,const int64_t*,const double*,int64_t,double*,hipStream_t){return hipSuccess;}
extern "C" hipError_t pdh_launch_moment(int,int,const PdhDev*,const double*,int,hipStream_t){return hipSuccess;}
extern "C" int pdh_moment_table_doubles(int n1d){const int NA=2*n1d-1;return 3*n1d*n1d*(NA+1)+2*n1d+NA*2*n1d+2*n1d*2*n1d;}
