// BPR hogwild on CPU with the negative drawn (a) uniformly over all items or (b) uniformly inside the positive's bin,
// bins re-dealt every epoch (items ranked by popularity, rank group g = rank / B dealt to the B bins by a random rotation)
#include <stdint.h>
#include <stdlib.h>
#include <math.h>
#include <omp.h>
static inline uint64_t sm64(uint64_t *s){ uint64_t z=(*s+=0x9E3779B97F4A7C15ull); z=(z^(z>>30))*0xBF58476D1CE4E5B9ull; z=(z^(z>>27))*0x94D049BB133111EBull; return z^(z>>31);}
static int has(const int32_t*ind,int lo,int hi,int c){int e=hi;while(lo<hi){int m=lo+((hi-lo)>>1); if(ind[m]<c)lo=m+1; else hi=m;} return lo<e&&ind[lo]==c;}
void sim_epochs(const int32_t*indptr,const int32_t*indices,const int32_t*user_ids,const int32_t*rank_item,const int32_t*item_rank,
                int64_t nnz,int n_items,float*U,float*V,float*Bi,int k,float lr,float reg,int nbins,int epochs,uint64_t seed,int epoch0){
  int n_groups=(n_items+nbins-1)/nbins;
  uint32_t*rot=malloc(sizeof(uint32_t)*n_groups);
  for(int e=0;e<epochs;++e){
    uint64_t s=seed*1315423911u+(uint64_t)(epoch0+e)*2654435761u; for(int g=0;g<n_groups;++g) rot[g]=(uint32_t)(sm64(&s)%(uint64_t)nbins);
#pragma omp parallel
    { uint64_t st=seed^((uint64_t)(omp_get_thread_num()+1)*0x9E3779B97F4A7C15ull)^((uint64_t)(epoch0+e)<<32);
#pragma omp for schedule(static)
      for(int64_t t=0;t<nnz;++t){
        int64_t ii=(int64_t)(sm64(&st)%(uint64_t)nnz); int u=user_ids[ii], i=indices[ii], j;
        if(nbins<=1){ j=(int)(sm64(&st)%(uint64_t)n_items);} else {
          int code=item_rank[i]; int g=code/nbins, slot=code%nbins; int p=(slot+rot[g])%nbins;
          // candidates: one item of every group whose slot maps to p
          int gj=(int)(sm64(&st)%(uint64_t)n_groups); int sj=((p-(int)rot[gj])%nbins+nbins)%nbins; int cj=gj*nbins+sj;
          if(cj>=n_items) continue; // partial last group: no item for p there (slight non-uniformity, fine for a sim)
          j=rank_item[cj]; }
        if(has(indices,indptr[u],indptr[u+1],j)) continue;
        float*pu=U+(size_t)u*k,*pi=V+(size_t)i*k,*pj=V+(size_t)j*k; float sc=Bi[i]-Bi[j];
        for(int f=0;f<k;++f) sc+=pu[f]*(pi[f]-pj[f]);
        float z=1.0f/(1.0f+expf(sc));
        for(int f=0;f<k;++f){ float tu=pu[f],ti=pi[f],tj=pj[f]; pu[f]+=lr*(z*(ti-tj)-reg*tu); pi[f]+=lr*(z*tu-reg*ti); pj[f]+=lr*(-z*tu-reg*tj);}
        Bi[i]+=lr*(z-reg*Bi[i]); Bi[j]+=lr*(-z-reg*Bi[j]);
      } }
  }
  free(rot);
}
