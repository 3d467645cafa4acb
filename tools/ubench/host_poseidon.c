#include <stdio.h>
#include <stdint.h>
#include <time.h>
#include <dlfcn.h>
int main(){ void* h=dlopen("leanmultisig_amd/libleanmultisig_hip.so",RTLD_NOW); if(!h){printf("%s\n",dlerror());return 1;}
 void (*f)(uint32_t*)=dlsym(h,"lmh_poseidon16_permute"); void (*g)(uint32_t*)=dlsym(h,"lmh_poseidon16_permute_scalar");
 uint32_t s[16]={1,2,3}; struct timespec a,b; int N=200000;
 for(int rep=0;rep<2;rep++){
 clock_gettime(CLOCK_MONOTONIC,&a); for(int i=0;i<N;i++) f(s); clock_gettime(CLOCK_MONOTONIC,&b);
 printf("avx512: %.3f us/perm (%u)\n",((b.tv_sec-a.tv_sec)*1e9+(b.tv_nsec-a.tv_nsec))/N/1e3,s[0]);
 clock_gettime(CLOCK_MONOTONIC,&a); for(int i=0;i<N;i++) g(s); clock_gettime(CLOCK_MONOTONIC,&b);
 printf("scalar: %.3f us/perm (%u)\n",((b.tv_sec-a.tv_sec)*1e9+(b.tv_nsec-a.tv_nsec))/N/1e3,s[0]);}
 return 0;}
