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
  // round 6, all-to-all merge (below): the slices publish tagged partial granules and finalise the blocks themselves
  unsigned long long* part_g = nullptr;  // [heads][64][HD + 2] {tag, fp32}; null = off
  const unsigned int* seq = nullptr;     // device-side step counter: tag = seq << 6 | layer
  int layer = 0;
  int* status = nullptr;                 // sticky give-up flag (bit 0)
};

// ---- all-to-all merge of context slices (round 6): no combine launch, no counter, no last arriver ---------------------
// Slice workgroup (h, s) publishes its partial — o[HD], max, sum — as {tag, fp32} granules [head][64 slices][HD + 2]
// (8-byte write-through agent-scope stores: the data is its own flag, cdna_hip_programming.md Guideline 16 form R2) and
// then FINALISES a share of the 16-value output blocks: one DPP row per block polls the ns slices' granules of its 16
// dims and runs attn_combine_kernel<128>'s sums in attn_combine_kernel's order (bit-identical to the combine launch).
// Two dependent trips (store, load) on the launch's tail instead of a 5-us launch; the three-trip counter form of round
// 4 (attn_slices_merge below) was slower than the launch. The slice workgroups of a head (group) wait for each other:
// the caller guarantees they can all become resident (grid <= the chip's slots, or beside workgroups that never wait).
struct AttnA2A {
  unsigned long long* part_g;
  unsigned int tag;
  int* status;
};
__device__ __forceinline__ unsigned long long* attn_part_granule(unsigned long long* g, int h, int s, int HD) {
  return g + ((size_t)h * ATTN_MAX_SLICES + s) * (HD + 2);
}
__device__ __forceinline__ void attn_a2a_publish(const AttnA2A& a, int h, int s, int HD, int d, float o) {
  __hip_atomic_store(attn_part_granule(a.part_g, h, s, HD) + d, ((unsigned long long)a.tag << 32) | __float_as_uint(o),
                     __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// Called by ALL 64 lanes of a wave; each DPP row (16 lanes) finalises block `b` of head `h` when `mine` (row-uniform).
// put(value, index into [heads * HD]) is called by whole rows. MAXS: compile-time bound of ns (registers: 2 per slice).
template <int HD, int MAXS, typename PUT>
__device__ __forceinline__ void attn_a2a_finalize(const AttnA2A& a, int h, int b, int ns, bool mine, const PUT& put) {
  static_assert(HD == 128, "the combine order reproduced here is attn_combine_kernel<128>'s (two thread groups)");
  const int lane = threadIdx.x & 63, j = lane & 15;
  const int dd = 16 * min(b, HD / 16 - 1) + j;
  unsigned long long go[MAXS], gm[(MAXS + 15) / 16], gl[(MAXS + 15) / 16];
  const unsigned long long t0 = wall_clock64();
  for (;;) {
    bool good = true;
#pragma unroll
    for (int s2 = 0; s2 < MAXS; ++s2) {
      go[s2] = __hip_atomic_load(attn_part_granule(a.part_g, h, min(s2, ns - 1), HD) + dd, __ATOMIC_RELAXED,
                                 __HIP_MEMORY_SCOPE_AGENT);
      good = good && (unsigned int)(go[s2] >> 32) == a.tag;
    }
#pragma unroll
    for (int q = 0; q < (MAXS + 15) / 16; ++q) {  // lane j holds the (max, sum) of slices j, j + 16, ...
      const unsigned long long* pm = attn_part_granule(a.part_g, h, min(j + 16 * q, ns - 1), HD) + HD;
      gm[q] = __hip_atomic_load(pm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      gl[q] = __hip_atomic_load(pm + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      good = good && (unsigned int)(gm[q] >> 32) == a.tag && (unsigned int)(gl[q] >> 32) == a.tag;
    }
    if (__all(good || !mine)) break;
    if (wall_clock64() - t0 > 2000000ull) {  // 20 ms: a slice of this head is missing — say so, do not hang
      if (lane == 0) atomicOr(a.status, 1);
      break;
    }
    __builtin_amdgcn_s_sleep(2);
  }
  // attn_combine_kernel's arithmetic, in its order: m = max_s m_s; w_s = e^(m_s - m); l = sum_s (l_s w_s) ascending;
  // o = (fma chain over even s ascending) + (fma chain over odd s ascending); result o / l
  float m_j[(MAXS + 15) / 16], m = -INFINITY;
#pragma unroll
  for (int q = 0; q < (MAXS + 15) / 16; ++q) {
    m_j[q] = j + 16 * q < ns ? __uint_as_float((unsigned int)gm[q]) : -INFINITY;
    m = fmaxf(m, m_j[q]);
  }
  m = row16_max(m);
  float w_j[(MAXS + 15) / 16], wl_j[(MAXS + 15) / 16];
#pragma unroll
  for (int q = 0; q < (MAXS + 15) / 16; ++q) {
    w_j[q] = m_j[q] == -INFINITY ? 0.f : __expf(m_j[q] - m);
    wl_j[q] = __uint_as_float((unsigned int)gl[q]) * w_j[q];
  }
  const int row0 = lane & ~15;
  float l = 0.f, oe = 0.f, oo = 0.f;
#pragma unroll
  for (int s2 = 0; s2 < MAXS; ++s2) {
    const float w_s = __shfl(w_j[s2 >> 4], row0 + (s2 & 15), 64), wl_s = __shfl(wl_j[s2 >> 4], row0 + (s2 & 15), 64);
    if (s2 < ns) {
      l += wl_s;
      const float v = __uint_as_float((unsigned int)go[s2]);
      if (s2 & 1)
        oo = fmaf(v, w_s, oo);
      else
        oe = fmaf(v, w_s, oe);
    }
  }
  if (mine) put((oe + oo) / l, h * HD + 16 * b + j);
}

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
