// woq_persist.h — host interface of the persistent token kernel (woq_persist.hip).
#pragma once
#include <string>

#include "woq_launch.h"
#include "woq_xq.h"

namespace woq {

struct Persist;  // device tables, vectors and the LDS plan of one engine

struct PersistDesc {
  int layers, hidden, inter, heads, kv_heads, head_dim, kv_dtype, max_ctx, window, attn_splits;
  float eps;
  const woq_layer_weights* lw;  // [layers]
  uint8_t* kcache;
  uint8_t* vcache;
  size_t kv_layer_bytes;
  const unsigned int* seq;  // device step counter (the embedding launch advances it)
  const int32_t* pos;
  int* status;
  const float* cs;
  const float* sn;
  XqPtrs x0;          // layer 0's input vector, written by the embedding launch
  const float* ssq0;  // its per-block sums of squares
  unsigned long long* qkv_g;
  float* hidden_buf;
};

Persist* persist_create(const PersistDesc& d, std::string* why);
void persist_destroy(Persist* p);
void persist_rebind(Persist* p, const int32_t* pos, float* hidden_buf);
int persist_ring_tiles(const Persist* p);
int persist_grid(const Persist* p);
// diagnostics: [grid][layers * 4][32] wall-clock (100 MHz) stamps of every projection (woq_persist.hip: PS_STAMP), null = off
void persist_set_stamps(Persist* p, unsigned long long* dev);
int persist_launch(Persist* p, hipStream_t st);

}  // namespace woq
