#include "../raytracer_amd/csrc/host_bvh.h"
#include <chrono>
#include <cstdio>
#include <random>
using namespace rt;
static double ha(const double* lo,const double* hi){double x=hi[0]-lo[0],y=hi[1]-lo[1],z=hi[2]-lo[2];return x*y+y*z+z*x;}
static void cost(const HostBvh& b,const char* name){
  double root=ha(&b.root_box.lo.x,&b.root_box.hi.x); double cn=0,cl=0;
  for(auto&n:b.nodes)for(int c=0;c<2;++c){double a=ha(n.lo[c],n.hi[c])/root; if(n.child[c]>=0)cn+=a; else cl+=a*(((~n.child[c])&1)+1);}
  printf("%s nodes=%zu depth=%d  E[node visits]=%.1f E[prim tests]=%.1f\n",name,b.nodes.size(),b.max_depth,cn+1,cl);
}
int main(){
  int n=1000000; std::mt19937_64 g(1); std::uniform_real_distribution<double> ux(-0.8,1.8),uy(-0.8,1.6),uz(-1.2,1.05),ur(0.00175,0.00525);
  std::vector<Box> boxes(n);
  for(auto&b:boxes){V3 c(ux(g),uy(g),uz(g)); V3 r(ur(g),ur(g),ur(g)); b=Box{c-r,c+r};}
  auto t0=std::chrono::steady_clock::now();
  HostBvh s=build_traversal_bvh(boxes,31);
  auto t1=std::chrono::steady_clock::now();
  printf("sah build %.2f s\n",std::chrono::duration<double>(t1-t0).count());
  cost(s,"sah");
  HostBvh s2=build_traversal_bvh(boxes,20);
  cost(s2,"balanced(depth20)");
}
