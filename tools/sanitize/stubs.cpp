// Stand-ins for the kernel launchers (the sanitizer build exercises the HOST half of the C ABI only: validation, run tables,
// eligibility tests of the row kernel - pdh_check_problem / pdh_check_rows / pdh_check_exchange; nothing is launched).
#include <hip/hip_runtime.h>
struct PdhDev;
struct PdhRows;
#define S(g) extern "C" hipError_t pdh_launch_g##g(int,int,int,int,int,const PdhDev*,int,size_t,hipStream_t){return hipSuccess;}
S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)
extern "C" hipError_t pdh_launch_rhs(int,int,const PdhDev*,int,const double*,const double*,double*,const int64_t*,const int64_t*,const int64_t*,hipStream_t){return hipSuccess;}
extern "C" hipError_t pdh_launch_eval(int,int,int,const PdhDev*,int,const double*,const int64_t*,const double*,int64_t,double*,double*,int,hipStream_t){return hipSuccess;}
extern "C" hipError_t pdh_launch_shape(int,int,const PdhDev*,int,const int64_t*,const double*,int64_t,double*,hipStream_t){return hipSuccess;}
extern "C" hipError_t pdh_launch_moment(int,int,const PdhDev*,const double*,int,hipStream_t){return hipSuccess;}
extern "C" hipError_t pdh_launch_rows(const PdhDev*,const PdhRows*,const double*,int,hipStream_t){return hipSuccess;}
extern "C" hipError_t pdh_launch_pack_faces(int,int64_t,const double*,const double*,const double*,const double*,int64_t,const int64_t*,const int64_t*,const int32_t*,const int32_t*,const double*,int64_t,double*,double*,double*,double*,double*,hipStream_t){return hipSuccess;}
extern "C" hipError_t pdh_launch_ghost_apply(const PdhDev*,const double*,int,const int64_t*,const int64_t*,const int32_t*,int,const int64_t*,const int64_t*,const int32_t*,hipStream_t){return hipSuccess;}
extern "C" hipError_t pdh_launch_checksum(const double*,int64_t,double*,hipStream_t){return hipSuccess;}
extern "C" int pdh_rows_max_faces(void){return 6;}
extern "C" int pdh_rows_n_dofs(int n1d,int basis){return basis ? n1d*(n1d+1)*(n1d+2)/6 : n1d*n1d*n1d;}
extern "C" int pdh_moment_table_doubles(int n1d){const int NA=2*n1d-1;return 3*n1d*n1d*(NA+1)+2*n1d+NA*2*n1d+2*n1d*2*n1d;}
