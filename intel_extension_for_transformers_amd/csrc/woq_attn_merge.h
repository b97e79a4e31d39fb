// woq_attn_merge.h — the merge of context-slice partials of the long-context decode attention, done by the LAST slice
// workgroup of a head (group) to finish instead of by a second launch (round 4).
//
// Why. attn_combine_kernel was a launch of `heads` workgroups that cost 5.2 us per layer at 8k context next to 11.2 us
// of slices (profiles/r01r_longctx_kernel_stats.txt): a launch floor (~1.7 us) plus two dependent rounds of loads of
// partials that another launch had just written. Here every slice workgroup publishes its partial with agent-scope
// write-through stores, drains them, and bumps a per-group counter; whoever sees the count complete reads all partials
// of the group (agent-scope loads, everything requested at once) and emits the attention output and its XQ form.
// Ordering is the counter's: an agent-scope atomic increment issued after `s_waitcnt vmcnt(0)` on write-through stores
// cannot be observed before those stores (cdna_hip_programming.md Guideline 16, form R1); the last arriver's loads
// bypass its XCD's L2 (sc1). The counter is left at zero for the next launch by the workgroup that completes it.
//
// Partial layout (fp32): o [head][64 slices][HD] un-normalised outputs, then ml [head][64][2] = {max (natural-log
// domain), sum}; 64 = the slice capacity of the buffer, whatever the launch's slice count.
#pragma once
#include "woq_device.h"
#include "woq_xq.h"

namespace woq {

constexpr int ATTN_MAX_SLICES = 64;

struct AttnMerge {
  unsigned int* counter;  // [groups], zero between launches; null = do not merge here (the combine launch does it)
  float* out;             // [heads][HD] attention output
  XqPtrs xo;              // its XQ form for the o_proj GEMV (or all null)
};

__device__ __forceinline__ float* attn_part_o(float* part, int h, int s, int HD) {
  return part + ((size_t)h * ATTN_MAX_SLICES + s) * HD;
}
__device__ __forceinline__ float* attn_part_ml(float* part, int heads, int h, int s, int HD) {
  return part + (size_t)heads * ATTN_MAX_SLICES * HD + ((size_t)h * ATTN_MAX_SLICES + s) * 2;
}
__device__ __forceinline__ void st_agent(float* p, float v) {
  __hip_atomic_store((unsigned int*)p, __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float ld_agent(const float* p) {
  return __uint_as_float(__hip_atomic_load((const unsigned int*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}

// Called by ALL 256 threads of a slice workgroup after its partials of heads [h0, h0 + nh) (slice `sp` of `ns`) went out
// through st_agent. `lds`: at least 3 * nh * ns + 256 + 8 floats of workgroup LDS nobody else is using any more.
// nh <= 8, ns <= 64, HD in {64, 128}.
template <int HD, int HB>  // HB = heads whose partial rows are in flight together (HB * 64 * HD / 256 registers per thread)
__device__ __forceinline__ void attn_slices_merge(float* part, int heads, int h0, int nh, int ns, unsigned int* counter,
                                                  const AttnMerge& mg, float* lds) {
  const int tid = threadIdx.x;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this thread's partial stores have reached the coherence point
  __syncthreads();
  unsigned int* flag = (unsigned int*)lds;
  if (tid == 0) {
    // acquire + release at agent scope: the arrival orders this workgroup's (already drained, write-through) partials
    // before it, and the last arriver's loads below after it — no reliance on in-order hardware or on the compiler
    // keeping ld_agent behind the barrier
    const unsigned int old = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    flag[0] = old == (unsigned int)(ns - 1) ? 1u : 0u;
    if (old == (unsigned int)(ns - 1))  // complete: nobody else touches it before the next launch
      __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  if (flag[0] == 0u) return;
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // every thread of the last arriver reads other workgroups' partials
  __syncthreads();  // flag word is about to be reused
  float* ms = lds;                // [nh][ns] slice maxima
  float* wl = ms + nh * ns;       // [nh][ns] slice sums -> weighted sums
  float* wsc = wl + nh * ns;      // [nh][ns] weights e^(m_s - m)
  float* hl = wsc + nh * ns;      // [nh] sum of the weighted sums
  float* red = hl + 8;            // [256]
  constexpr int GROUPS = 256 / HD;  // thread groups that walk alternate slices
  constexpr int PER = ATTN_MAX_SLICES / GROUPS;  // partial rows one thread may have to fetch per head
  const int d = tid % HD, grp = tid / HD;
  // ONE round trip: the maxima / sums and the first batch of partial rows are all requested — branch-free, into
  // registers — before anything is waited for. (First version: head by head and slice group by slice group, ten
  // dependent trips to memory, 8.7 us on the launch's tail against 5.2 us for a whole combine launch; second version:
  // predicated loads behind a load -> LDS-store loop, still ~10 us — profiles/r04b_longctx_ab.txt, r04c_*.) Rows past
  // ns are fetched from row ns - 1 and weighted 0.
  float m_r[2], l_r[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int i = min(tid + 256 * j, nh * ns - 1);
    const int hh = i / ns, s = i - hh * ns;
    const float* p = attn_part_ml(part, heads, h0 + hh, s, HD);
    m_r[j] = ld_agent(p);
    l_r[j] = ld_agent(p + 1);
  }
  for (int hb = 0; hb < nh; hb += HB) {
    float v[HB][PER];
#pragma unroll
    for (int j = 0; j < HB; ++j) {
      const float* row0 = attn_part_o(part, h0 + min(hb + j, nh - 1), 0, HD) + d;
#pragma unroll
      for (int u = 0; u < PER; ++u) v[j][u] = ld_agent(row0 + (size_t)min(grp + u * GROUPS, ns - 1) * HD);
    }
    if (hb == 0) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
        if (tid + 256 * j < nh * ns) ms[tid + 256 * j] = m_r[j], wl[tid + 256 * j] = l_r[j];
    }
    if (hb == 0) {  // weights of every head of the group (LDS only; the partial rows above are still in flight)
      __syncthreads();
      for (int i = tid; i < nh * ns; i += 256) {
        const int hh = i / ns;
        float m = -INFINITY;
        for (int s = 0; s < ns; ++s) m = fmaxf(m, ms[hh * ns + s]);
        const float w = ms[i] == -INFINITY ? 0.f : __expf(ms[i] - m);
        wsc[i] = w;
        wl[i] *= w;  // own element only
      }
      __syncthreads();
      if (tid < nh) {
        float l = 0.f;
        for (int s = 0; s < ns; ++s) l += wl[tid * ns + s];
        hl[tid] = l;
      }
    }
#pragma unroll
    for (int j = 0; j < HB; ++j) {
      if (hb + j >= nh) break;  // (uniform)
      const int hh = hb + j;
      float o = 0.f;
#pragma unroll
      for (int u = 0; u < PER; ++u) {
        const int s = grp + u * GROUPS;
        if (s < ns) o = fmaf(v[j][u], wsc[hh * ns + s], o);  // ascending s: the combine launch's order
      }
      __syncthreads();  // red is free again (and hl / wsc are settled on the first pass)
      red[tid] = o;
      __syncthreads();
      if (grp == 0) {
        float t = 0.f;
#pragma unroll
        for (int g2 = 0; g2 < GROUPS; ++g2) t += red[g2 * HD + d];
        const float r = t / hl[hh];
        const int idx = (h0 + hh) * HD + d;
        mg.out[idx] = r;
        if (mg.xo.limbs != nullptr) xq_emit16(r, mg.xo, idx >> 4, d & 15);
      }
    }
  }
}

}  // namespace woq
