// Wall time of mppi_numpy_normals over thread counts (host only): g++ -O3 -std=c++17 -ffp-contract=off tools/numpy_stream_time.cpp -lpthread
#include "../trajopt_amd/csrc/numpy_stream.cpp"
#include <chrono>
#include <cstdio>
int main(int argc,char**argv){
  size_t n=4096*64*7; std::vector<double> out(n); uint32_t st[628];
  for(int nt: {1,2,4,8,16}){
    double best=1e9;
    for(int r=0;r<9;r++){
      auto t0=std::chrono::steady_clock::now();
      mppi_numpy_normals(123,n,out.data(),nt,st);
      auto t1=std::chrono::steady_clock::now();
      double ms=std::chrono::duration<double,std::milli>(t1-t0).count(); if(ms<best)best=ms;
    }
    printf("threads %d: %.3f ms\n",nt,best);
  }
}
