/*
 * gnnflow_oracle.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C, single-threaded CPU restatement of the reference's temporal edge
 * store and sampler (jasperzhong/GNNFlow @ /root/reference).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it, and only
 * as the checker / the reported CPU baseline.  The product path
 * (gnnflow_amd/csrc) never links or calls this file.
 *
 * It deliberately keeps the reference's *data structure* (per-node doubly
 * linked list of time-sorted blocks, newest at the tail) and *per-(root, slot)
 * algorithm* (block walk + 4-way case split + LowerBound), so that parity of the
 * HIP path (which uses a flattened per-node segment layout) against this oracle
 * demonstrates the layout-equivalence claim of DESIGN.md rather than assuming it.
 *
 * Parity pinning: checked against the literal expectations of the reference's
 * own tests (tests/test_temporal_sampler.py, tests/test_dynamic_graph.py),
 * transcribed as data into tests/golden/reference_tests.json
 * (see tests/test_oracle_golden.py).  The reference's native build is
 * unbuildable here (CUDA + rmm fork + thrust), so there is no
 * oracle/_ref.
 *
 * Each function cites the reference lines it restates.
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/gnnflow_rng.h"

#define GFO_OK 0
#define GFO_ERR_ARG 1      /* bad argument (negative node id, size mismatch) */
#define GFO_ERR_ORDER 2    /* CHECK_LE(block->end_timestamp, last new ts) failed */
#define GFO_ERR_NOMEM 3

/* ---- common.h:13-24,35-48 --------------------------------------------------- */
typedef int64_t nid_t;
typedef int64_t eid_t;
typedef float ts_t;
#define K_INVALID_NID (-1)
#define K_BLOCK_SPACE 20u /* sizeof(NID)+sizeof(EID)+sizeof(TS), common.h:23-24 */

typedef struct Block {
  nid_t* dst;
  ts_t* ts;
  eid_t* eid;
  size_t size, capacity;
  ts_t start_ts, end_ts;
  struct Block *prev, *next;
} Block;

/* doubly_linked_list.h:23-36 (host mirror; the device one keeps only tail) */
typedef struct {
  Block *head, *tail;
  size_t num_edges, num_insertions, size;
} List;

/* tiny open-addressing map eid -> multiplicity (dynamic_graph.h:153
 * std::unordered_map<EIDType, std::size_t> edges_) */
typedef struct {
  eid_t* keys;
  size_t* vals;
  uint8_t* used;
  size_t cap, nused, nlive;
} EidMap;

typedef struct gfo_graph {
  List* table;
  size_t table_len; /* == max_node_id + 1 once any node was added */
  size_t max_node_id;
  uint8_t* node_seen; /* nodes_ (std::set) as a bitmap over [0, table_len) */
  uint8_t* src_seen;  /* src_nodes_ */
  size_t num_nodes, num_src_nodes;
  EidMap edges;
  size_t min_block_size;
  int policy; /* 0 = insert, 1 = replace (common.h:75) */
  int adaptive;
  size_t allocated_bytes; /* temporal_block_allocator.cu:148 allocated_ */
  size_t num_blocks;      /* h2d_mapping_.size() */
} gfo_graph;

/* ---- eid map ---------------------------------------------------------------- */
static uint64_t mix64(uint64_t x) {
  x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ULL;
  x ^= x >> 27; x *= 0x94d049bb133111ebULL;
  x ^= x >> 31;
  return x;
}
static int eidmap_grow(EidMap* m) {
  size_t ncap = m->cap ? m->cap * 2 : 1024;
  eid_t* k = (eid_t*)malloc(ncap * sizeof(eid_t));
  size_t* v = (size_t*)malloc(ncap * sizeof(size_t));
  uint8_t* u = (uint8_t*)calloc(ncap, 1);
  if (!k || !v || !u) return GFO_ERR_NOMEM;
  for (size_t i = 0; i < m->cap; ++i) {
    if (!m->used[i]) continue;
    size_t p = mix64((uint64_t)m->keys[i]) & (ncap - 1);
    while (u[p]) p = (p + 1) & (ncap - 1);
    u[p] = 1; k[p] = m->keys[i]; v[p] = m->vals[i];
  }
  free(m->keys); free(m->vals); free(m->used);
  m->keys = k; m->vals = v; m->used = u; m->cap = ncap;
  return GFO_OK;
}
static size_t* eidmap_slot(EidMap* m, eid_t key, int create) {
  if (create && (m->nused + 1) * 2 > m->cap) {
    if (eidmap_grow(m) != GFO_OK) return NULL;
  }
  if (!m->cap) return NULL;
  size_t p = mix64((uint64_t)key) & (m->cap - 1);
  while (m->used[p]) {
    if (m->keys[p] == key) return &m->vals[p];
    p = (p + 1) & (m->cap - 1);
  }
  if (!create) return NULL;
  m->used[p] = 1; m->keys[p] = key; m->vals[p] = 0; m->nused++;
  return &m->vals[p];
}

/* ---- allocator: temporal_block_allocator.cu:83-88,134-176 ------------------- */
static size_t align_up(const gfo_graph* g, size_t size) {
  return size < g->min_block_size ? g->min_block_size : size;
}
static int block_alloc_internal(gfo_graph* g, Block* b, size_t size) {
  size_t cap = align_up(g, size);
  b->size = 0;
  b->capacity = cap;
  b->start_ts = FLT_MAX; /* std::numeric_limits<TimestampType>::max() */
  b->end_ts = 0;
  b->prev = b->next = NULL;
  b->dst = (nid_t*)malloc((cap ? cap : 1) * sizeof(nid_t));
  b->ts = (ts_t*)malloc((cap ? cap : 1) * sizeof(ts_t));
  b->eid = (eid_t*)malloc((cap ? cap : 1) * sizeof(eid_t));
  if (!b->dst || !b->ts || !b->eid) return GFO_ERR_NOMEM;
  g->allocated_bytes += cap * K_BLOCK_SPACE;
  return GFO_OK;
}
static void block_free_internal(gfo_graph* g, Block* b) {
  free(b->dst); free(b->ts); free(b->eid);
  b->dst = NULL; b->ts = NULL; b->eid = NULL;
  g->allocated_bytes -= b->capacity * K_BLOCK_SPACE;
  b->size = 0;
  b->capacity = 0;
}

/* ---- utils.cu:33-63 CopyEdgesToBlock ---------------------------------------- */
static int copy_edges_to_block(Block* b, const nid_t* dst, const ts_t* ts,
                               const eid_t* eid, size_t start_idx, size_t n) {
  if (b->size + n > b->capacity) return GFO_ERR_ARG;
  /* "we assume that the incoming edges are newer than the existing ones" */
  if (!(b->end_ts <= ts[start_idx + n - 1])) return GFO_ERR_ORDER;
  memcpy(b->dst + b->size, dst + start_idx, n * sizeof(nid_t));
  memcpy(b->ts + b->size, ts + start_idx, n * sizeof(ts_t));
  memcpy(b->eid + b->size, eid + start_idx, n * sizeof(eid_t));
  b->size += n;
  b->start_ts = b->start_ts < ts[start_idx] ? b->start_ts : ts[start_idx];
  b->end_ts = ts[start_idx + n - 1];
  return GFO_OK;
}

/* ---- doubly_linked_list.cu:45-85 (host versions) ---------------------------- */
static void list_insert(List* l, Block* b) {
  if (l->tail == NULL) {
    l->tail = l->head = b;
    b->prev = b->next = NULL;
  } else {
    l->tail->next = b;
    b->prev = l->tail;
    b->next = NULL;
    l->tail = b;
  }
  l->size++;
}
static void list_remove(List* l, Block* b) {
  if (b->prev == NULL && b->next == NULL) {
    l->head = l->tail = NULL;
  } else if (b->prev == NULL) {
    l->head = b->next;
    b->next->prev = NULL;
  } else if (b->next == NULL) {
    l->tail = b->prev;
    b->prev->next = NULL;
  } else {
    b->prev->next = b->next;
    b->next->prev = b->prev;
  }
  l->size--;
}

/* dynamic_graph.cu:202-204 get_next_power_of_two (n >= 1; n == 1 -> 1) */
static size_t next_pow2(size_t n) {
  size_t p = 1;
  while (p < n) p <<= 1;
  return p;
}

/* ---- graph lifetime --------------------------------------------------------- */
gfo_graph* gfo_graph_create(size_t min_block_size, int insertion_policy,
                            int adaptive_block_size) {
  gfo_graph* g = (gfo_graph*)calloc(1, sizeof(gfo_graph));
  if (!g) return NULL;
  g->min_block_size = min_block_size;
  g->policy = insertion_policy;
  g->adaptive = adaptive_block_size;
  return g;
}

void gfo_graph_destroy(gfo_graph* g) {
  if (!g) return;
  for (size_t i = 0; i < g->table_len; ++i) {
    Block* b = g->table[i].head;
    while (b) {
      Block* n = b->next;
      free(b->dst); free(b->ts); free(b->eid);
      free(b);
      b = n;
    }
  }
  free(g->table); free(g->node_seen); free(g->src_seen);
  free(g->edges.keys); free(g->edges.vals); free(g->edges.used);
  free(g);
}

/* dynamic_graph.cu:140-147 AddNodes */
static int add_nodes(gfo_graph* g, nid_t max_node) {
  if (g->table_len > 0 && (size_t)max_node < g->max_node_id) return GFO_OK;
  size_t nlen = (size_t)max_node + 1;
  if (nlen > g->table_len) {
    List* t = (List*)realloc(g->table, nlen * sizeof(List));
    uint8_t* ns = (uint8_t*)realloc(g->node_seen, nlen);
    uint8_t* ss = (uint8_t*)realloc(g->src_seen, nlen);
    if (t) g->table = t;
    if (ns) g->node_seen = ns;
    if (ss) g->src_seen = ss;
    if (!t || !ns || !ss) return GFO_ERR_NOMEM;
    memset(g->table + g->table_len, 0, (nlen - g->table_len) * sizeof(List));
    memset(g->node_seen + g->table_len, 0, nlen - g->table_len);
    memset(g->src_seen + g->table_len, 0, nlen - g->table_len);
    g->table_len = nlen;
  }
  g->max_node_id = (size_t)max_node;
  return GFO_OK;
}

/* ---- dynamic_graph.cu:206-287 AddEdgesForOneNode ---------------------------- */
static int add_edges_for_one_node(gfo_graph* g, nid_t src, const nid_t* dst,
                                  const ts_t* ts, const eid_t* eid, size_t n) {
  size_t num_edges = n;
  List* list = &g->table[src];
  Block* tail = list->tail;
  Block* blk = NULL;
  int is_new = 0, rc;
  size_t start_idx = 0;

  if (tail == NULL) {
    /* case 1: empty list */
    blk = (Block*)calloc(1, sizeof(Block));
    if (!blk) return GFO_ERR_NOMEM;
    if ((rc = block_alloc_internal(g, blk, num_edges))) return rc;
    is_new = 1;
  } else if (tail->size + num_edges > tail->capacity) {
    /* case 2: not enough space in the current block */
    if (g->policy == 0) {
      size_t fill = tail->capacity - tail->size;
      if (fill > 0) {
        if ((rc = copy_edges_to_block(tail, dst, ts, eid, 0, fill))) return rc;
        start_idx = fill;
        num_edges -= fill;
      }
      size_t avg = list->num_insertions == 0
                       ? num_edges
                       : list->num_edges / list->num_insertions;
      size_t new_size;
      if (g->adaptive) {
        new_size = num_edges > avg ? num_edges : avg;
        new_size = next_pow2(new_size);
      } else {
        new_size = num_edges;
      }
      blk = (Block*)calloc(1, sizeof(Block));
      if (!blk) return GFO_ERR_NOMEM;
      if ((rc = block_alloc_internal(g, blk, new_size))) return rc;
      is_new = 1;
    } else {
      /* replace: temporal_block_allocator.cu:122-132 Reallocate */
      Block tmp;
      if ((rc = block_alloc_internal(g, &tmp, tail->size + num_edges))) return rc;
      memcpy(tmp.dst, tail->dst, tail->size * sizeof(nid_t));
      memcpy(tmp.ts, tail->ts, tail->size * sizeof(ts_t));
      memcpy(tmp.eid, tail->eid, tail->size * sizeof(eid_t));
      tmp.size = tail->size;
      tmp.start_ts = tail->start_ts;
      tmp.end_ts = tail->end_ts;
      tmp.next = tail->next;
      /* the reference's `*block = tmp` drops `prev` (AllocateInternal nulls it);
       * with REPLACE a list never has more than one block, so prev is NULL */
      tmp.prev = tail->prev;
      block_free_internal(g, tail);
      *tail = tmp;
    }
  }
  if (!is_new) blk = tail; /* case 3 */

  if ((rc = copy_edges_to_block(blk, dst, ts, eid, start_idx, num_edges))) {
    if (is_new) { block_free_internal(g, blk); free(blk); }
    return rc;
  }
  if (is_new) {
    list_insert(list, blk);
    g->num_blocks++;
  }
  list->num_edges += n;
  list->num_insertions++;
  return GFO_OK;
}

/* sort key for "group by src, then stable sort by timestamp"
 * (dynamic_graph.cu:105-128, utils.h:16-27): order by (src, ts, input index) */
typedef struct { nid_t src; ts_t ts; size_t idx; } SortKey;
static int sortkey_cmp(const void* a, const void* b) {
  const SortKey* x = (const SortKey*)a;
  const SortKey* y = (const SortKey*)b;
  if (x->src != y->src) return x->src < y->src ? -1 : 1;
  if (x->ts < y->ts) return -1;
  if (y->ts < x->ts) return 1;
  return x->idx < y->idx ? -1 : (x->idx > y->idx);
}

/* ---- dynamic_graph.cu:77-138 AddEdges --------------------------------------- */
int gfo_graph_add_edges(gfo_graph* g, const nid_t* src, const nid_t* dst,
                        const ts_t* ts, const eid_t* eid, size_t n) {
  if (n == 0) return GFO_ERR_ARG; /* CHECK_GT(src_nodes.size(), 0) */
  nid_t max_node = 0;
  for (size_t i = 0; i < n; ++i) {
    if (src[i] < 0 || dst[i] < 0) return GFO_ERR_ARG;
    if (src[i] > max_node) max_node = src[i];
    if (dst[i] > max_node) max_node = dst[i];
  }
  int rc = add_nodes(g, max_node);
  if (rc) return rc;
  for (size_t i = 0; i < n; ++i) {
    if (!g->src_seen[src[i]]) { g->src_seen[src[i]] = 1; g->num_src_nodes++; }
    if (!g->node_seen[src[i]]) { g->node_seen[src[i]] = 1; g->num_nodes++; }
    if (!g->node_seen[dst[i]]) { g->node_seen[dst[i]] = 1; g->num_nodes++; }
    size_t* c = eidmap_slot(&g->edges, eid[i], 1);
    if (!c) return GFO_ERR_NOMEM;
    if (*c == 0) g->edges.nlive++;
    (*c)++;
  }
  SortKey* keys = (SortKey*)malloc(n * sizeof(SortKey));
  nid_t* sdst = (nid_t*)malloc(n * sizeof(nid_t));
  ts_t* sts = (ts_t*)malloc(n * sizeof(ts_t));
  eid_t* seid = (eid_t*)malloc(n * sizeof(eid_t));
  if (!keys || !sdst || !sts || !seid) {
    free(keys); free(sdst); free(sts); free(seid);
    return GFO_ERR_NOMEM;
  }
  for (size_t i = 0; i < n; ++i) {
    keys[i].src = src[i]; keys[i].ts = ts[i]; keys[i].idx = i;
  }
  qsort(keys, n, sizeof(SortKey), sortkey_cmp);
  for (size_t i = 0; i < n; ++i) {
    sdst[i] = dst[keys[i].idx];
    sts[i] = ts[keys[i].idx];
    seid[i] = eid[keys[i].idx];
  }
  rc = GFO_OK;
  for (size_t i = 0; i < n && rc == GFO_OK;) {
    size_t j = i;
    while (j < n && keys[j].src == keys[i].src) ++j;
    rc = add_edges_for_one_node(g, keys[i].src, sdst + i, sts + i, seid + i, j - i);
    i = j;
  }
  free(keys); free(sdst); free(sts); free(seid);
  return rc;
}

/* ---- dynamic_graph.cu:382-411 OffloadOldBlocks ------------------------------- */
size_t gfo_graph_offload_old_blocks(gfo_graph* g, ts_t timestamp) {
  size_t num = 0;
  for (size_t node = 0; node < g->table_len; ++node) {
    if (!g->node_seen[node]) continue;
    List* list = &g->table[node];
    Block* cur = list->head; /* oldest block */
    while (cur) {
      Block* next = cur->next;
      if (cur->end_ts < timestamp) {
        for (size_t i = 0; i < cur->size; ++i) {
          size_t* c = eidmap_slot(&g->edges, cur->eid[i], 0);
          if (c && *c > 0 && --(*c) == 0) g->edges.nlive--;
        }
        list_remove(list, cur);
        g->num_blocks--;
        block_free_internal(g, cur);
        free(cur);
        num++;
      }
      cur = next;
    }
  }
  return num;
}

/* ---- accessors: dynamic_graph.cu:149-151,289-380 ---------------------------- */
size_t gfo_graph_num_nodes(const gfo_graph* g) { return g->num_nodes; }
size_t gfo_graph_num_src_nodes(const gfo_graph* g) { return g->num_src_nodes; }
size_t gfo_graph_num_edges(const gfo_graph* g) { return g->edges.nlive; }
int64_t gfo_graph_max_node_id(const gfo_graph* g) { return (int64_t)g->max_node_id; }

int gfo_graph_out_degree(const gfo_graph* g, const nid_t* nodes, size_t n,
                         size_t* out) {
  for (size_t i = 0; i < n; ++i) {
    if (nodes[i] < 0 || (size_t)nodes[i] >= g->table_len) return GFO_ERR_ARG;
    out[i] = g->table[nodes[i]].num_edges;
  }
  return GFO_OK;
}

size_t gfo_graph_nodes(const gfo_graph* g, nid_t* out, int src_only) {
  size_t k = 0;
  const uint8_t* seen = src_only ? g->src_seen : g->node_seen;
  for (size_t i = 0; i < g->table_len; ++i)
    if (seen[i]) { if (out) out[k] = (nid_t)i; k++; }
  return k;
}

size_t gfo_graph_edges(const gfo_graph* g, eid_t* out) {
  size_t k = 0;
  for (size_t i = 0; i < g->edges.cap; ++i)
    if (g->edges.used[i] && g->edges.vals[i] > 0) { if (out) out[k] = g->edges.keys[i]; k++; }
  return k;
}

/* newest first: tail -> head, each block reversed (dynamic_graph.cu:300-337) */
size_t gfo_graph_get_temporal_neighbors(const gfo_graph* g, nid_t node,
                                        nid_t* dst, ts_t* ts, eid_t* eid) {
  if (node < 0 || (size_t)node >= g->table_len) return 0;
  size_t k = 0;
  for (const Block* b = g->table[node].tail; b; b = b->prev) {
    for (size_t i = b->size; i-- > 0;) {
      if (dst) { dst[k] = b->dst[i]; ts[k] = b->ts[i]; eid[k] = b->eid[i]; }
      k++;
    }
  }
  return k;
}

float gfo_graph_avg_linked_list_length(const gfo_graph* g) {
  float sum = 0;
  for (size_t i = 0; i < g->table_len; ++i)
    if (g->node_seen[i]) sum += (float)g->table[i].size;
  return sum / (float)g->num_nodes;
}
float gfo_graph_mem_usage(const gfo_graph* g) { return (float)g->allocated_bytes; }
float gfo_graph_metadata_mem_usage(const gfo_graph* g) {
  /* sizeof(TemporalBlock) = 64, sizeof(DoublyLinkedList) = 8 */
  return (float)(64 * g->num_blocks + 8 * g->table_len);
}

/* block introspection (tests of the block policy) */
size_t gfo_graph_num_blocks(const gfo_graph* g, nid_t node) {
  if (node < 0 || (size_t)node >= g->table_len) return 0;
  return g->table[node].size;
}
/* idx 0 = oldest (head) */
int gfo_graph_block_info(const gfo_graph* g, nid_t node, size_t idx, size_t* size,
                         size_t* capacity, ts_t* start_ts, ts_t* end_ts) {
  if (node < 0 || (size_t)node >= g->table_len) return GFO_ERR_ARG;
  const Block* b = g->table[node].head;
  while (b && idx--) b = b->next;
  if (!b) return GFO_ERR_ARG;
  *size = b->size; *capacity = b->capacity;
  *start_ts = b->start_ts; *end_ts = b->end_ts;
  return GFO_OK;
}

/* ---- utils.cu:96-109 LowerBound ---------------------------------------------- */
static int lower_bound_ts(const ts_t* ts, int n, ts_t x) {
  int left = 0, right = n;
  while (left < right) {
    int mid = (left + right) / 2;
    if (ts[mid] < x) left = mid + 1; else right = mid;
  }
  return left;
}

/* sampling_kernels.cu:28-40 — time window of (root, snapshot).
 * `end - k*w` is written with fmaf because nvcc contracts it (fmad is on by
 * default and the build adds --use_fast_math, CMakeLists.txt:53); the two agree
 * whenever k*w is exactly representable (every config in gnnflow/config.py). */
static void time_window(ts_t root_ts, uint32_t snapshot_idx, uint32_t num_snapshots,
                        ts_t window, ts_t* start, ts_t* end) {
  if (num_snapshots == 1) {
    *start = (fabs((double)window) < 1e-6) ? 0.0f : root_ts - window;
    *end = root_ts;
  } else {
    ts_t k = (ts_t)(num_snapshots - snapshot_idx - 1);
    *end = fmaf(-k, window, root_ts);
    *start = *end - window;
  }
}

/* sampling_kernels.cu:55-86 — candidate index range of one block */
static void block_range(const Block* b, ts_t start, ts_t end, int* s, int* e) {
  if (start >= b->start_ts && end <= b->end_ts) {
    *s = lower_bound_ts(b->ts, (int)b->size, start);
    *e = lower_bound_ts(b->ts, (int)b->size, end);
  } else if (start < b->start_ts && end <= b->end_ts) {
    *s = 0;
    *e = lower_bound_ts(b->ts, (int)b->size, end);
  } else if (start > b->start_ts && end > b->end_ts) {
    *s = lower_bound_ts(b->ts, (int)b->size, start);
    *e = (int)b->size;
  } else {
    *s = 0;
    *e = (int)b->size;
  }
}

/* One (root, slot) "thread" of SampleLayerRecentKernel / SampleLayerUniformKernel
 * (sampling_kernels.cu:11-107, :109-273).  Returns 1 and fills the outputs if a
 * neighbour was selected, 0 for an invalid slot (src = kInvalidNID). */
static int sample_slot(const gfo_graph* g, int uniform, nid_t nid, ts_t start, ts_t end,
                       uint32_t sample_index, uint64_t seed, uint64_t tid,
                       uint64_t call, nid_t* o_dst, eid_t* o_eid, ts_t* o_ts) {
  if (nid < 0 || (size_t)nid >= g->table_len) return 0; /* ref: out-of-bounds read */
  const List* list = &g->table[nid];
  int64_t index;
  if (uniform) {
    /* pass 1 (:143-200): count candidates over the walked blocks */
    uint32_t num_candidates = 0;
    for (const Block* b = list->tail; b; b = b->prev) {
      if (b->capacity == 0) continue;
      if (end < b->start_ts) continue;
      if (start > b->end_ts) break;
      int s, e;
      block_range(b, start, end, &s, &e);
      num_candidates += (uint32_t)(e - s);
    }
    /* :202 `curand(...) % num_candidates`; 0 candidates is a modulo by zero in
     * the reference and ends in an invalid slot — defined as invalid here */
    if (num_candidates == 0) return 0;
    index = gf_philox4x32_10_first(seed, tid, call) % num_candidates;
  } else {
    index = sample_index;
  }
  /* selection walk (:43-106 / :205-272), newest block first */
  for (const Block* b = list->tail; b; b = b->prev) {
    if (b->capacity == 0) continue;
    if (end < b->start_ts) continue;
    if (start > b->end_ts) break;
    int s, e;
    block_range(b, start, end, &s, &e);
    int64_t i = (int64_t)e - 1 - index;
    if (i < s) {
      index -= e - s;
      continue;
    }
    *o_dst = b->dst[i];
    *o_eid = b->eid[i];
    *o_ts = b->ts[i];
    return 1;
  }
  return 0;
}

/*
 * TemporalSampler::SampleLayer (temporal_sampler.cu:97-277): kernel over R*F
 * slots, stable compaction of valid slots (thrust::remove_if, :191-204), result
 * assembly (:236-274).
 *
 * Outputs (caller allocates): all_nodes[R + R*F], all_ts[R + R*F], dt[R*F],
 * eids[R*F], row[R*F], col[R*F]; *num_sampled = S.  num_dst = R, num_src = R+S.
 * policy: 0 recent, 1 uniform.  `call` = index of this sample_layer invocation
 * on the sampler (uniform RNG counter, see include/gnnflow_rng.h).
 */
int gfo_sample_layer(const gfo_graph* g, int policy, uint32_t fanout,
                     uint32_t num_snapshots, uint32_t snapshot_idx, ts_t window,
                     int prop_time, uint64_t seed, uint64_t call, const nid_t* roots,
                     const ts_t* root_ts, size_t R, nid_t* all_nodes, ts_t* all_ts,
                     ts_t* dt, eid_t* eids, nid_t* row, nid_t* col,
                     size_t* num_sampled) {
  size_t S = 0;
  for (size_t r = 0; r < R; ++r) {
    all_nodes[r] = roots[r];
    all_ts[r] = root_ts[r];
  }
  for (size_t r = 0; r < R; ++r) {
    ts_t start, end;
    time_window(root_ts[r], snapshot_idx, num_snapshots, window, &start, &end);
    for (uint32_t j = 0; j < fanout; ++j) {
      nid_t d; eid_t e; ts_t t;
      uint64_t tid = (uint64_t)r * fanout + j;
      if (!sample_slot(g, policy == 1, roots[r], start, end, j, seed, tid, call,
                       &d, &e, &t))
        continue;
      all_nodes[R + S] = d;
      all_ts[R + S] = prop_time ? root_ts[r] : t;
      dt[S] = root_ts[r] - t;
      eids[S] = e;
      row[S] = (nid_t)r;
      col[S] = (nid_t)(R + S);
      S++;
    }
  }
  *num_sampled = S;
  return GFO_OK;
}

/*
 * The same layer on several host threads (OpenMP), for bench.py's cpu_baseline only: roots
 * are independent, so pass 1 fills fixed slots per root in parallel with the SAME per-slot
 * routine (sample_slot) and pass 2 is the same stable compaction, serial.  Outputs are
 * identical to gfo_sample_layer (tests/test_oracle_golden.py checks that).
 */
int gfo_sample_layer_mt(const gfo_graph* g, int policy, uint32_t fanout,
                        uint32_t num_snapshots, uint32_t snapshot_idx, ts_t window,
                        int prop_time, uint64_t seed, uint64_t call, const nid_t* roots,
                        const ts_t* root_ts, size_t R, nid_t* all_nodes, ts_t* all_ts,
                        ts_t* dt, eid_t* eids, nid_t* row, nid_t* col,
                        size_t* num_sampled, int threads) {
  size_t slots = R * (size_t)fanout;
  nid_t* sd = (nid_t*)malloc((slots ? slots : 1) * sizeof(nid_t));
  eid_t* se = (eid_t*)malloc((slots ? slots : 1) * sizeof(eid_t));
  ts_t* st = (ts_t*)malloc((slots ? slots : 1) * sizeof(ts_t));
  if (!sd || !se || !st) { free(sd); free(se); free(st); return GFO_ERR_NOMEM; }
  long long r;
#pragma omp parallel for schedule(dynamic, 64) num_threads(threads > 0 ? threads : 1)
  for (r = 0; r < (long long)R; ++r) {
    ts_t start, end;
    time_window(root_ts[r], snapshot_idx, num_snapshots, window, &start, &end);
    for (uint32_t j = 0; j < fanout; ++j) {
      uint64_t tid = (uint64_t)r * fanout + j;
      nid_t d; eid_t e; ts_t t;
      if (sample_slot(g, policy == 1, roots[r], start, end, j, seed, tid, call, &d, &e, &t)) {
        sd[tid] = d; se[tid] = e; st[tid] = t;
      } else {
        sd[tid] = -1;
      }
    }
  }
  size_t S = 0;
  for (size_t i = 0; i < R; ++i) {
    all_nodes[i] = roots[i];
    all_ts[i] = root_ts[i];
  }
  for (size_t i = 0; i < R; ++i) {
    for (uint32_t j = 0; j < fanout; ++j) {
      size_t tid = i * fanout + j;
      if (sd[tid] < 0) continue;
      all_nodes[R + S] = sd[tid];
      all_ts[R + S] = prop_time ? root_ts[i] : st[tid];
      dt[S] = root_ts[i] - st[tid];
      eids[S] = se[tid];
      row[S] = (nid_t)i;
      col[S] = (nid_t)(R + S);
      S++;
    }
  }
  free(sd); free(se); free(st);
  *num_sampled = S;
  return GFO_OK;
}

int gfo_gather_rows_mt(const float* feats, size_t num_rows, size_t dim,
                       const int64_t* ids, size_t n, float* out, int threads) {
  int bad = 0;
  long long i;
#pragma omp parallel for schedule(static) num_threads(threads > 0 ? threads : 1)
  for (i = 0; i < (long long)n; ++i) {
    if (ids[i] < 0 || (size_t)ids[i] >= num_rows) { bad = 1; continue; }
    memcpy(out + (size_t)i * dim, feats + (size_t)ids[i] * dim, dim * sizeof(float));
  }
  return bad ? GFO_ERR_ARG : GFO_OK;
}

/* Cache-free feature gather, gnnflow/utils.py:465-474 prepare_input:
 * out[i, :] = feats[ids[i], :] (float32 rows). */
int gfo_gather_rows(const float* feats, size_t num_rows, size_t dim,
                    const int64_t* ids, size_t n, float* out) {
  for (size_t i = 0; i < n; ++i) {
    if (ids[i] < 0 || (size_t)ids[i] >= num_rows) return GFO_ERR_ARG;
    memcpy(out + i * dim, feats + (size_t)ids[i] * dim, dim * sizeof(float));
  }
  return GFO_OK;
}

uint32_t gfo_philox_first(uint64_t seed, uint64_t tid, uint64_t call) {
  return gf_philox4x32_10_first(seed, tid, call);
}
