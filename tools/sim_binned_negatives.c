// BPR hogwild on CPU (the reference's arithmetic, recom_bpr.pyx:231-267) with the negative drawn (a) uniformly over all
// items or (b) uniformly among the other items of the positive's bin; the bins of an epoch are given by the caller
// (bin_of_item + the bins' item lists), i.e. by the restated deal of the HIP path (oracle.ldsbin_deal_key).
#include <stdint.h>
#include <stdlib.h>
#include <math.h>
#include <omp.h>
static inline uint64_t sm64(uint64_t *s){ uint64_t z=(*s+=0x9E3779B97F4A7C15ull); z=(z^(z>>30))*0xBF58476D1CE4E5B9ull; z=(z^(z>>27))*0x94D049BB133111EBull; return z^(z>>31);}
static int has(const int32_t*ind,int lo,int hi,int c){int e=hi;while(lo<hi){int m=lo+((hi-lo)>>1); if(ind[m]<c)lo=m+1; else hi=m;} return lo<e&&ind[lo]==c;}
void sim_epoch(const int32_t*indptr,const int32_t*indices,const int32_t*user_ids,const int32_t*bin_of_item,const int32_t*bptr,
               const int32_t*bitems,int64_t nnz,int n_items,float*U,float*V,float*Bi,int k,float lr,float reg,int nbins,uint64_t seed,int epoch){
#pragma omp parallel
    { uint64_t st=seed^((uint64_t)(omp_get_thread_num()+1)*0x9E3779B97F4A7C15ull)^((uint64_t)epoch<<32);
#pragma omp for schedule(static)
      for(int64_t t=0;t<nnz;++t){
        int64_t ii=(int64_t)(sm64(&st)%(uint64_t)nnz); int u=user_ids[ii], i=indices[ii], j;
        if(nbins<=1){ j=(int)(sm64(&st)%(uint64_t)n_items);} else {
          int b=bin_of_item[i]; int n=bptr[b+1]-bptr[b]; if(n<2) continue;
          int s=(int)(sm64(&st)%(uint64_t)(n-1)); j=bitems[bptr[b]+s]; if(j==i) j=bitems[bptr[b]+n-1]; }
        if(has(indices,indptr[u],indptr[u+1],j)) continue;
        float*pu=U+(size_t)u*k,*pi=V+(size_t)i*k,*pj=V+(size_t)j*k; float sc=Bi[i]-Bi[j];
        for(int f=0;f<k;++f) sc+=pu[f]*(pi[f]-pj[f]);
        float z=1.0f/(1.0f+expf(sc));
        for(int f=0;f<k;++f){ float tu=pu[f],ti=pi[f],tj=pj[f]; pu[f]+=lr*(z*(ti-tj)-reg*tu); pi[f]+=lr*(z*tu-reg*ti); pj[f]+=lr*(-z*tu-reg*tj);}
        Bi[i]+=lr*(z-reg*Bi[i]); Bi[j]+=lr*(-z-reg*Bi[j]);
      } }
}
