// Development aid: the host codec tiers (AVX-512 blocks vs SSE groups) side by side, ns per call by size.
//   g++ -O2 -std=c++17 -o /tmp/hcp tools/native/host_codec_paths.cpp && /tmp/hcp
#include "../../streamvbyte_amd/csrc/svb_cpu.h"
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace svbcpu;
static double now(){return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();}
volatile size_t sink;
template<class F> double timeit(F f){ unsigned reps=1; for(;;){double t0=now(); for(unsigned r=0;r<reps;++r) f(); double dt=now()-t0; if(dt>0.004){reps=(unsigned)(reps*0.02/dt)+1;break;} reps*=4;} double best=1e30; for(int k=0;k<7;++k){double t0=now(); for(unsigned r=0;r<reps;++r) f(); double dt=(now()-t0)/reps; if(dt<best)best=dt;} return best*1e9;}
int main(){ const uint32_t N=65535; std::vector<uint32_t> v(N+16), back(N+16); std::vector<uint8_t> buf(5*N+64); srand(1); for(uint32_t k=0;k<N;++k) v[k]=(uint32_t)rand()>>(31u&(uint32_t)rand());
 for(uint32_t n: {4u,16u,32u,64u,128u,256u,1024u,65535u}){
  uint8_t* keys; uint8_t* data; uint32_t prev=0;
  double e_full=timeit([&]{sink=encode<k1234,false>(v.data(),n,buf.data(),0);});
  double d_full=timeit([&]{sink=decode<k1234,false>(buf.data(),back.data(),n,0);});
  // SSE only
  double e_sse=timeit([&]{ uint8_t* k=buf.data(); uint8_t* d=buf.data()+(n+3)/4; uint32_t p=0; uint32_t done=0; if(n>=8) done=encode_groups_sse41<k1234,false>(v.data(),n,k,d,p); d=encode_scalar_from<k1234,false>(v.data(),done,n,k,d,p); sink=d-buf.data();});
  encode<k1234,false>(v.data(),n,buf.data(),0);
  double d_sse=timeit([&]{ const uint8_t* k=buf.data(); const uint8_t* d=buf.data()+(n+3)/4; uint32_t p=0; uint32_t done=0; if(n>=8) done=decode_groups_ssse3<k1234,false>(k,d,back.data(),n,p); d=decode_scalar_from<k1234,false>(k,d,back.data(),done,n,p); sink=d-buf.data();});
  printf("n=%6u encode full %8.1f sse %8.1f | decode full %8.1f sse %8.1f ns\n",n,e_full,e_sse,d_full,d_sse);
 }
}
