// Experiment (not shipped): how sequential is the reference flood?  For the exact pop sequence,
// a "round" is a maximal run of consecutive pops that were all already in the heap when the round
// began (age <= age counter at round start) -- the ideal commit length of a speculative batch.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef struct { float v; int64_t age; int64_t idx; } el;
static inline int less(const el*a,const el*b){ if(a->v<b->v)return 1; if(a->v>b->v)return 0; if(a->age!=b->age)return a->age<b->age; return a->idx<b->idx;}
int64_t flood_stats(const float*aff,const int64_t*seeds,int64_t ns,const uint8_t*mask,uint32_t*out,const int64_t*dims,int64_t*hist,int nh,int64_t*maxheap){
  int64_t N=dims[0]*dims[1]*dims[2],SZ=dims[1]*dims[2],SY=dims[2];
  const int64_t noff[6]={-SZ,-SY,-1,1,SY,SZ}; const int ax[6]={0,1,2,2,1,0};
  for(int64_t i=0;i<ns;++i) out[seeds[i]]=(uint32_t)(i+1);
  int64_t cap=ns+16,hn=0; el*h=malloc(sizeof(el)*cap);
  #define PUSH(E) do{ if(hn==cap){cap*=2;h=realloc(h,sizeof(el)*cap);} int64_t c=hn++; while(c>0){int64_t p=(c-1)>>1; if(!less(&(E),&h[p]))break; h[c]=h[p]; c=p;} h[c]=(E);}while(0)
  int64_t age=0,pops=0,rounds=0,round_age=0,run=0,mh=0;
  for(int64_t i=0;i<ns;++i){el e={0.f,0,seeds[i]};PUSH(e);}
  while(hn>0){
    if(hn>mh)mh=hn;
    el top=h[0]; el last=h[--hn];
    if(hn>0){int64_t c=0;for(;;){int64_t l=2*c+1,r=l+1,m;if(l>=hn)break;m=(r<hn&&less(&h[r],&h[l]))?r:l;if(!less(&h[m],&last))break;h[c]=h[m];c=m;}h[c]=last;}
    ++pops;
    if(top.age>round_age){ // pushed during this round: round ends
      int b=0; int64_t r=run; while(r>1&&b<nh-1){r>>=1;++b;} hist[b]+=1; ++rounds; run=0; round_age=age; }
    ++run;
    for(int k=0;k<6;++k){int64_t nb=top.idx+noff[k]; if(!mask[nb]||out[nb])continue; out[nb]=out[top.idx]; int64_t src=(k<3)?top.idx:nb; el e={aff[(int64_t)ax[k]*N+src],++age,nb}; PUSH(e);}
  }
  free(h); *maxheap=mh; return rounds;
}
