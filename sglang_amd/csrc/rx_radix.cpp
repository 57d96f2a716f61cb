// Native radix tree of cached KV prefixes (host side of RadixAttention).
//
// Contract = the reference's RadixCache (python/sglang/srt/mem_cache/radix_cache.py:279-812):
// match_prefix (:352-410) returns the concatenated KV slot indices of the longest page-aligned
// cached prefix and splits a node when the match ends inside it (:674-694); insert (:412-432,
// :704-757); lock refs (:592-626) move tokens between evictable and protected; evict (:562-590)
// pops evictable leaves from a heap ordered by the eviction policy
// (python/sglang/srt/mem_cache/evict_policy.py) and frees their slots.  The reference also ships a
// C++ variant (srt/mem_cache/cpp_radix_tree/); this one is written from the Python contract.
//
// Differences by design: node keys/values are int64 vectors in host memory (values are KV slot
// ids; the caller uploads the matched run once), "time" is a logical tick (one per tree
// operation, like the reference's single time.monotonic() per helper call) with node id as the
// deterministic tie-break (a node created during an operation gets its own later tick, as
// TreeNode.__init__ does), and nodes are addressed by id across the C ABI.
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <map>
#include <queue>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "../../include/radix_hip.h"

namespace {

struct Node;
using ChildKey = std::pair<std::string, std::vector<int64_t>>;  // (extra_key, first page of tokens)

struct Node {
  int64_t id = 0;
  Node* parent = nullptr;
  std::string extra;
  std::vector<int64_t> key;
  std::vector<int64_t> value;
  std::map<ChildKey, Node*> children;
  int32_t lock_ref = 0;
  int32_t hit_count = 0;
  int64_t priority = 0;
  uint64_t last_access = 0;
  uint64_t creation = 0;
};

enum Policy { LRU = 0, LFU = 1, FIFO = 2, MRU = 3, FILO = 4, PRIORITY = 5, SLRU = 6 };

struct PrioKey {
  int64_t a;
  int64_t b;
  int64_t id;
  bool operator>(const PrioKey& o) const {
    if (a != o.a) return a > o.a;
    if (b != o.b) return b > o.b;
    return id > o.id;
  }
};

}  // namespace

struct rx_radix {
  int page_size = 1;
  int policy = LRU;
  Node* root = nullptr;
  int64_t next_id = 0;
  uint64_t tick = 0;
  int64_t evictable_size = 0;
  int64_t protected_size = 0;
  std::unordered_map<int64_t, Node*> by_id;
  std::unordered_set<Node*> evictable_leaves;

  Node* new_node(int64_t priority) {
    Node* n = new Node();
    n->id = next_id++;
    n->priority = priority;
    n->last_access = n->creation = ++tick;  // TreeNode() stamps time.monotonic() at creation,
                                            // i.e. later than the operation's access time
    by_id[n->id] = n;
    return n;
  }
  void free_subtree(Node* n) {
    for (auto& kv : n->children) free_subtree(kv.second);
    by_id.erase(n->id);
    delete n;
  }
  void reset() {
    if (root) free_subtree(root);
    by_id.clear();
    evictable_leaves.clear();
    evictable_size = protected_size = 0;
    root = new_node(INT64_MIN);
    root->lock_ref = 1;  // radix_cache.py:335
  }
  ChildKey child_key(const std::string& extra, const int64_t* t) const {
    return ChildKey(extra, std::vector<int64_t>(t, t + page_size));
  }
  // RadixKey.match (:171-212): common prefix length rounded down to the page size
  int64_t match_len(const std::vector<int64_t>& a, const int64_t* b, int64_t nb) const {
    const int64_t n = std::min<int64_t>(a.size(), nb);
    int64_t i = 0;
    while (i < n && a[i] == b[i]) ++i;
    return page_size == 1 ? i : (i / page_size) * page_size;
  }
  void update_leaf_status(Node* n) {  // :783-797 (no host tier: a node is never "evicted")
    if (n->lock_ref > 0) {
      evictable_leaves.erase(n);
      return;
    }
    if (!n->children.empty()) {
      evictable_leaves.erase(n);
      return;
    }
    evictable_leaves.insert(n);
  }
  Node* split(Node* child, int64_t split_len) {  // :674-694
    Node* nn = new_node(child->priority);
    nn->hit_count = child->hit_count;
    nn->extra = child->extra;
    nn->parent = child->parent;
    nn->lock_ref = child->lock_ref;
    const ChildKey old_key = child_key(child->extra, child->key.data());
    nn->key.assign(child->key.begin(), child->key.begin() + split_len);
    nn->value.assign(child->value.begin(), child->value.begin() + split_len);
    child->key.erase(child->key.begin(), child->key.begin() + split_len);
    child->value.erase(child->value.begin(), child->value.begin() + split_len);
    nn->children[child_key(child->extra, child->key.data())] = child;
    child->parent = nn;
    nn->parent->children[old_key] = nn;
    return nn;
  }
  PrioKey prio(const Node* n) const {  // evict_policy.py
    const int64_t la = static_cast<int64_t>(n->last_access), cr = static_cast<int64_t>(n->creation);
    switch (policy) {
      case LFU: return {n->hit_count, la, n->id};
      case FIFO: return {cr, 0, n->id};
      case MRU: return {-la, 0, n->id};
      case FILO: return {-cr, 0, n->id};
      case PRIORITY: return {n->priority, la, n->id};
      case SLRU: return {n->hit_count >= 2 ? 1 : 0, la, n->id};
      default: return {la, 0, n->id};
    }
  }
};

extern "C" {

rx_radix* rx_radix_create(int page_size, int eviction_policy) {
  if (page_size < 1 || eviction_policy < 0 || eviction_policy > SLRU) return nullptr;
  rx_radix* t = new rx_radix();
  t->page_size = page_size;
  t->policy = eviction_policy;
  t->reset();
  return t;
}

void rx_radix_destroy(rx_radix* t) {
  if (!t) return;
  if (t->root) t->free_subtree(t->root);
  delete t;
}

void rx_radix_reset(rx_radix* t) { t->reset(); }

int64_t rx_radix_root(const rx_radix* t) { return t->root->id; }

int64_t rx_radix_match_prefix(rx_radix* t, const int64_t* token_ids, int64_t n, const char* extra_key,
                              int64_t* out_indices, int64_t cap, int64_t* last_node) {
  const std::string extra = extra_key ? extra_key : "";
  n = (n / t->page_size) * t->page_size;  // key.page_aligned (:400)
  Node* node = t->root;
  if (last_node) *last_node = node->id;
  if (n <= 0) return 0;
  const uint64_t now = ++t->tick;
  node->last_access = now;
  int64_t done = 0;
  while (done < n) {
    auto it = node->children.find(t->child_key(extra, token_ids + done));
    if (it == node->children.end()) break;
    Node* child = it->second;
    child->last_access = now;
    const int64_t pl = t->match_len(child->key, token_ids + done, n - done);
    Node* taken = child;
    if (pl < static_cast<int64_t>(child->key.size())) taken = t->split(child, pl);
    if (done + pl > cap) return -1;
    if (out_indices) std::memcpy(out_indices + done, taken->value.data(), pl * sizeof(int64_t));
    done += pl;
    node = taken;
    if (taken != child) break;  // matched inside a node: stop after the split (:659-663)
  }
  if (last_node) *last_node = node->id;
  return done;
}

int64_t rx_radix_insert(rx_radix* t, const int64_t* token_ids, const int64_t* values, int64_t n,
                        const char* extra_key, int64_t priority, int chunked, int64_t* last_node) {
  const std::string extra = extra_key ? extra_key : "";
  n = (n / t->page_size) * t->page_size;
  Node* node = t->root;
  const uint64_t now = ++t->tick;
  node->last_access = now;
  node->priority = std::max(node->priority, priority);
  int64_t done = 0, total_prefix = 0;
  while (done < n) {
    auto it = node->children.find(t->child_key(extra, token_ids + done));
    if (it == node->children.end()) break;
    node = it->second;
    node->last_access = now;
    const int64_t pl = t->match_len(node->key, token_ids + done, n - done);
    total_prefix += pl;
    done += pl;
    if (pl < static_cast<int64_t>(node->key.size())) node = t->split(node, pl);
    node->priority = std::max(node->priority, priority);
    if (!chunked) node->hit_count += 1;
  }
  if (done < n) {
    Node* nn = t->new_node(priority);
    nn->parent = node;
    nn->extra = extra;
    nn->key.assign(token_ids + done, token_ids + n);
    nn->value.assign(values + done, values + n);
    if (!chunked) nn->hit_count += 1;
    node->children[t->child_key(extra, token_ids + done)] = nn;
    t->evictable_size += n - done;
    t->update_leaf_status(node);
    t->update_leaf_status(nn);
    node = nn;
  }
  if (last_node) *last_node = node->id;
  return total_prefix;
}

int64_t rx_radix_inc_lock_ref(rx_radix* t, int64_t node_id) {  // :592-605
  auto it = t->by_id.find(node_id);
  if (it == t->by_id.end()) return INT64_MIN;
  int64_t delta = 0;
  for (Node* n = it->second; n != t->root; n = n->parent) {
    if (n->lock_ref == 0) {
      const int64_t len = n->key.size();
      t->evictable_size -= len;
      t->protected_size += len;
      delta -= len;
    }
    n->lock_ref += 1;
    t->update_leaf_status(n);
  }
  return delta;
}

int64_t rx_radix_dec_lock_ref(rx_radix* t, int64_t node_id) {  // :607-626
  auto it = t->by_id.find(node_id);
  if (it == t->by_id.end()) return INT64_MIN;
  int64_t delta = 0;
  for (Node* n = it->second; n != t->root; n = n->parent) {
    if (n->lock_ref == 1) {
      const int64_t len = n->key.size();
      t->evictable_size += len;
      t->protected_size -= len;
      delta += len;
    }
    n->lock_ref -= 1;
    t->update_leaf_status(n);
  }
  return delta;
}

// Evicts leaves in policy order until >= num_tokens slots are freed (:562-590).  The freed slot
// ids are written node by node into out_slots, and each node's slot count into out_seg_lens, so
// the caller can hand every segment to the allocator in the same order as the reference
// (free_segment(x.value, start_pos=0) per evicted node).
int64_t rx_radix_evict(rx_radix* t, int64_t num_tokens, int64_t* out_slots, int64_t slot_cap,
                       int64_t* out_seg_lens, int64_t seg_cap, int64_t* num_segments) {
  using Item = std::pair<PrioKey, Node*>;
  auto cmp = [](const Item& a, const Item& b) { return a.first > b.first; };
  std::priority_queue<Item, std::vector<Item>, decltype(cmp)> heap(cmp);
  for (Node* n : t->evictable_leaves) heap.push({t->prio(n), n});
  int64_t evicted = 0, segs = 0;
  while (evicted < num_tokens && !heap.empty()) {
    Node* x = heap.top().second;
    heap.pop();
    const int64_t len = x->value.size();
    if (evicted + len > slot_cap || segs >= seg_cap) break;
    std::memcpy(out_slots + evicted, x->value.data(), len * sizeof(int64_t));
    out_seg_lens[segs++] = len;
    evicted += len;
    Node* parent = x->parent;
    parent->children.erase(t->child_key(x->extra, x->key.data()));  // _delete_leaf (:771-781)
    t->evictable_size -= x->key.size();
    t->evictable_leaves.erase(x);
    t->by_id.erase(x->id);
    delete x;
    t->update_leaf_status(parent);
    if (parent->children.empty() && parent->lock_ref == 0) heap.push({t->prio(parent), parent});
  }
  if (num_segments) *num_segments = segs;
  return evicted;
}

int64_t rx_radix_evictable_size(const rx_radix* t) { return t->evictable_size; }
int64_t rx_radix_protected_size(const rx_radix* t) { return t->protected_size; }

int64_t rx_radix_total_size(const rx_radix* t) {
  int64_t total = 0;
  std::vector<const Node*> stack{t->root};
  while (!stack.empty()) {
    const Node* n = stack.back();
    stack.pop_back();
    total += n->value.size();
    for (auto& kv : n->children) stack.push_back(kv.second);
  }
  return total;
}

int64_t rx_radix_num_nodes(const rx_radix* t) { return static_cast<int64_t>(t->by_id.size()); }

// Node introspection for tests / debugging: fills {parent id, key length, lock_ref, hit_count,
// number of children, priority}; returns 0, or -1 for an unknown id.
int rx_radix_node_info(const rx_radix* t, int64_t node_id, int64_t* info6) {
  auto it = t->by_id.find(node_id);
  if (it == t->by_id.end()) return -1;
  const Node* n = it->second;
  info6[0] = n->parent ? n->parent->id : -1;
  info6[1] = n->key.size();
  info6[2] = n->lock_ref;
  info6[3] = n->hit_count;
  info6[4] = n->children.size();
  info6[5] = n->priority;
  return 0;
}

// One request's cache bookkeeping in ONE call: what RadixCache.cache_finished_req (:434-486) and
// cache_unfinished_req (:488-553) spread over insert / match_prefix / dec_lock_ref / inc_lock_ref plus the slot-range
// arithmetic between them.  The caller passes the request's tokens and the KV slots of its req_to_token row (host
// copies); the tree is updated and the caller gets back WHICH PARTS OF THE ROW to hand to the allocator and (for an
// unfinished request) the row's new contents.
//   flags: bit 0 finished, bit 1 insert (finished only: 0 = free everything past the protected prefix),
//          bit 2 chunked (unfinished only)
//   out8:  [0],[1] first range to free  [begin, end) as offsets into the row  (duplicates of pages the tree
//                  already held, or the whole uninserted run)           -- free_segment(start_pos = begin)
//          [2],[3] second range (finished only): the tail past the last whole page [key_len, n)
//          [4]     length of the page-aligned key
//          [5]     unfinished: number of slots written to out_slots (the row's cached prefix as the tree now
//                  holds it -- possibly pages shared with an earlier request)
//          [6]     unfinished: id of the node the request is now locked on
//          [7]     prefix length the insert found already cached
// last_node: the node the request held a lock on (< 0: none); it is released here, and for an unfinished request
// the new last node is locked.  Returns 0, or -1 on bad arguments / a too-small out_slots.
int rx_radix_cache_req(rx_radix* t, const int64_t* token_ids, const int64_t* slots, int64_t n, const char* extra_key,
                       int priority, int flags, int64_t protected_len, int64_t last_node, int64_t* out_slots,
                       int64_t out_cap, int64_t* out8) {
  if (!t || !out8 || n < 0 || (n > 0 && (!token_ids || !slots)) || protected_len < 0) return -1;
  const bool finished = flags & 1, do_insert = flags & 2, chunked = flags & 4;
  const int64_t key_len = n / t->page_size * t->page_size;
  for (int i = 0; i < 8; ++i) out8[i] = 0;
  out8[4] = key_len;
  int64_t node_id = t->root->id;
  if (finished) {
    int64_t freed_end = key_len;
    if (do_insert) {
      freed_end = rx_radix_insert(t, token_ids, slots, key_len, extra_key, priority, 0, &node_id);
      out8[7] = freed_end;
    }
    out8[0] = protected_len;
    out8[1] = freed_end > protected_len ? freed_end : protected_len;
    out8[2] = key_len;
    out8[3] = n;
    if (last_node >= 0) rx_radix_dec_lock_ref(t, last_node);
    return 0;
  }
  const int64_t pre = rx_radix_insert(t, token_ids, slots, key_len, extra_key, priority, chunked ? 1 : 0, &node_id);
  out8[7] = pre;
  out8[0] = protected_len;
  out8[1] = pre > protected_len ? pre : protected_len;
  if (out_cap < key_len || (key_len > 0 && !out_slots)) return -1;
  int64_t new_last = t->root->id;
  const int64_t m = key_len ? rx_radix_match_prefix(t, token_ids, key_len, extra_key, out_slots, out_cap, &new_last) : 0;
  if (m != key_len) return -1;  // everything just inserted must match (radix_cache.py:527)
  out8[5] = m;
  out8[6] = new_last;
  if (last_node >= 0) rx_radix_dec_lock_ref(t, last_node);
  rx_radix_inc_lock_ref(t, new_last);
  return 0;
}

}  // extern "C"
