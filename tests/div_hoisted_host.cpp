// Bit-for-bit comparison of inflx_div_by_hoisted (csrc/inflx_device_math.h) with the IEEE division it
// replaces, on the host -- TEST INFRASTRUCTURE.  usage: div_hoisted_host [millions of random quotients]
#include <cmath>
#include <cstdlib>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <random>
#define INFLX_HOST_TWIN 1
#define INFLX_FN static inline
using std::sqrt; using std::sin; using std::cos; using std::exp; using std::log; using std::fabs; using std::tanh; using std::cosh; using std::sinh; using std::tan;
#include "inflx_device_math.h"
static uint64_t bits(double x){uint64_t u; memcpy(&u,&x,8); return u;}
int main(int argc, char** argv){
  const long millions = argc > 1 ? atol(argv[1]) : 200;
  std::mt19937_64 rng(12345);
  long bad=0, n=0, slow=0;
  auto check=[&](double a,double b){
    const double y=inflx_recip(b); bool ok=true; double q=inflx_div_by_hoisted(a,b,y,ok); const double t=a/b; ++n;
    if(!ok) { q=t; ++slow; }  // the generated point stage re-evaluates such a point with IEEE divisions
    if(!(bits(q)==bits(t) || (q!=q && t!=t))) { if(bad<10) printf("MISMATCH a=%a b=%a got=%a want=%a\n",a,b,q,t); ++bad; }
  };
  // random significands, moderate exponents
  for(long i=0;i<millions*1000000L;++i){
    uint64_t ma=rng(), mb=rng();
    double a,b; uint64_t ua=(ma&0x800FFFFFFFFFFFFFull)|((uint64_t)(1023+(int)((ma>>52)&0x3F)-32)<<52), ub=(mb&0x800FFFFFFFFFFFFFull)|((uint64_t)(1023+(int)((mb>>52)&0x3F)-32)<<52);
    memcpy(&a,&ua,8); memcpy(&b,&ub,8); check(a,b);
  }
  printf("random: %ld quotients, %ld mismatches\n",n,bad); printf("  (of which %ld took the slow path)\n",slow); if(slow>n/1000) { printf("too many slow-path cases\n"); return 2; }
  // full exponent range incl. denormals / overflow
  n=0; long bad0=bad;
  for(long i=0;i<millions*100000L;++i){ uint64_t ua=rng(), ub=rng(); double a,b; memcpy(&a,&ua,8); memcpy(&b,&ub,8); check(a,b);} 
  printf("all bit patterns: %ld quotients, %ld mismatches\n",n,bad-bad0);
  // specials and hard cases
  const double sp[]={0.0,-0.0,1.0,-1.0,INFINITY,-INFINITY,NAN,5e-324,2.2250738585072014e-308,1.7976931348623157e308,1e-310,3.0,1.0/3.0,0x1.fffffffffffffp0,0x1.0000000000001p0,0x1.fffffffffffffp-1};
  n=0; bad0=bad;
  for(double a:sp) for(double b:sp) check(a,b);
  // significands near all-ones and near powers of two
  for(int i=0;i<2000000;++i){ uint64_t k=rng()%64, l=rng()%64; double a=std::ldexp((double)((1ull<<53)-1-k),(int)(rng()%40)-20-52), b=std::ldexp((double)((1ull<<53)-1-l),(int)(rng()%40)-20-52); check(a,b); check(b,a); double c=std::ldexp((double)((1ull<<52)+k),-52); check(a,c); check(c,b);} 
  printf("specials + hard cases: %ld quotients, %ld mismatches\n",n,bad-bad0);
  return bad?1:0;
}
