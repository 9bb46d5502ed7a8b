// mpt.cpp -- see mpt.hpp.  Keccak-256, RLP, and the partial Merkle-Patricia trie of the decoder rows.
#include "mpt.hpp"

namespace mpt {

const H256 EMPTY_TRIE_HASH = {86, 232, 31, 23, 27, 204, 85, 166, 255, 131, 69, 230, 146, 192, 248, 110,
                              91, 72, 224, 27, 153, 108, 173, 192, 1, 98, 47, 181, 227, 99, 180, 33};
const H256 EMPTY_CODE_HASH = {197, 210, 70, 1, 134, 247, 35, 60, 146, 126, 125, 178, 220, 199, 3, 192,
                              229, 0, 182, 83, 202, 130, 39, 59, 123, 250, 216, 4, 93, 133, 164, 112};

// ---------------------------------------------------------------- Keccak-256 (FIPS 202 permutation, 0x01 padding)
static void keccak_f(uint64_t st[25]) {
  static const uint64_t RC[24] = {
      0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL,
      0x000000000000808bULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL,
      0x000000000000008aULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000aULL,
      0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
      0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
      0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
  static const int ROT[24] = {1, 3, 6, 10, 15, 21, 28, 36, 45, 55, 2, 14, 27, 41, 56, 8, 25, 43, 62, 18, 39, 61, 20, 44};
  static const int PIL[24] = {10, 7, 11, 17, 18, 3, 5, 16, 8, 21, 24, 4, 15, 23, 19, 13, 12, 2, 20, 14, 22, 9, 6, 1};
  for (int r = 0; r < 24; r++) {
    uint64_t bc[5];
    for (int i = 0; i < 5; i++) bc[i] = st[i] ^ st[i + 5] ^ st[i + 10] ^ st[i + 15] ^ st[i + 20];
    for (int i = 0; i < 5; i++) {
      uint64_t t = bc[(i + 4) % 5] ^ ((bc[(i + 1) % 5] << 1) | (bc[(i + 1) % 5] >> 63));
      for (int j = 0; j < 25; j += 5) st[j + i] ^= t;
    }
    uint64_t t = st[1];
    for (int i = 0; i < 24; i++) {
      int j = PIL[i];
      uint64_t b = st[j];
      st[j] = (t << ROT[i]) | (t >> (64 - ROT[i]));
      t = b;
    }
    for (int j = 0; j < 25; j += 5) {
      uint64_t row[5];
      for (int i = 0; i < 5; i++) row[i] = st[j + i];
      for (int i = 0; i < 5; i++) st[j + i] ^= (~row[(i + 1) % 5]) & row[(i + 2) % 5];
    }
    st[0] ^= RC[r];
  }
}
H256 keccak256(const uint8_t* data, size_t len) { return keccak256_traced(data, len, nullptr); }

// The absorbing side of the same hash, block by block, as rows of the Keccak sponge table (air.hpp, AIR 6): per block
// 44 words = flags (1 full block, 2 final block), message bytes in the block, the block as absorbed (17 words, pad10*1
// included on the final one), the 25 lanes of the state BEFORE the block.
H256 keccak256_sponge_rows(const uint8_t* data, size_t len, std::vector<uint64_t>* rows) {
  uint64_t st[25] = {0};
  const size_t rate = 136;
  for (;;) {
    const bool final_block = len < rate;
    uint8_t block[136];
    std::memset(block, 0, rate);
    const size_t take = final_block ? len : rate;
    if (take) std::memcpy(block, data, take);
    if (final_block) {
      block[len] ^= 0x01;
      block[rate - 1] ^= 0x80;
    }
    uint64_t w[17];
    std::memcpy(w, block, rate);
    if (rows) {
      rows->push_back(final_block ? 2 : 1);
      rows->push_back(take);
      rows->insert(rows->end(), w, w + 17);
      rows->insert(rows->end(), st, st + 25);
    }
    for (size_t i = 0; i < 17; i++) st[i] ^= w[i];
    keccak_f(st);
    if (final_block) break;
    data += rate;
    len -= rate;
  }
  H256 out;
  std::memcpy(out.data(), st, 32);
  return out;
}

// The same sponge; every state that goes INTO a permutation (25 lanes, index x + 5y) is appended to `perm_inputs`:
// the rows a Keccak-f table needs to attest this hash (one permutation per 136-byte block, padding included).
H256 keccak256_traced(const uint8_t* data, size_t len, std::vector<uint64_t>* perm_inputs) {
  uint64_t st[25] = {0};
  const size_t rate = 136;
  uint8_t block[136];
  while (len >= rate) {
    for (size_t i = 0; i < rate / 8; i++) {
      uint64_t w;
      std::memcpy(&w, data + 8 * i, 8);
      st[i] ^= w;
    }
    if (perm_inputs) perm_inputs->insert(perm_inputs->end(), st, st + 25);
    keccak_f(st);
    data += rate;
    len -= rate;
  }
  std::memset(block, 0, rate);
  if (len) std::memcpy(block, data, len);
  block[len] ^= 0x01;
  block[rate - 1] ^= 0x80;
  for (size_t i = 0; i < rate / 8; i++) {
    uint64_t w;
    std::memcpy(&w, block + 8 * i, 8);
    st[i] ^= w;
  }
  if (perm_inputs) perm_inputs->insert(perm_inputs->end(), st, st + 25);
  keccak_f(st);
  H256 out;
  std::memcpy(out.data(), st, 32);
  return out;
}

// ---------------------------------------------------------------- RLP
static void rlp_len_prefix(Bytes& out, size_t len, uint8_t short_base) {
  if (len < 56) {
    out.push_back((uint8_t)(short_base + len));
  } else {
    uint8_t tmp[8];
    int n = 0;
    for (size_t l = len; l; l >>= 8) tmp[n++] = (uint8_t)l;
    out.push_back((uint8_t)(short_base + 55 + n));
    for (int i = n - 1; i >= 0; i--) out.push_back(tmp[i]);
  }
}
Bytes rlp_string(const uint8_t* d, size_t len) {
  Bytes out;
  if (len == 1 && d[0] < 0x80) {
    out.push_back(d[0]);
    return out;
  }
  rlp_len_prefix(out, len, 0x80);
  out.insert(out.end(), d, d + len);
  return out;
}
Bytes rlp_list(const std::vector<Bytes>& items) {
  size_t total = 0;
  for (auto& i : items) total += i.size();
  Bytes out;
  rlp_len_prefix(out, total, 0xc0);
  for (auto& i : items) out.insert(out.end(), i.begin(), i.end());
  return out;
}
Bytes rlp_scalar_be(const Bytes& be) {
  size_t i = 0;
  while (i < be.size() && be[i] == 0) i++;
  return rlp_string(be.data() + i, be.size() - i);
}
Bytes rlp_u64(uint64_t v) {
  Bytes be(8);
  for (int i = 0; i < 8; i++) be[i] = (uint8_t)(v >> (56 - 8 * i));
  return rlp_scalar_be(be);
}
bool rlp_parse(const uint8_t* d, size_t n, RlpItem* out) {
  if (n == 0) return false;
  const uint8_t b = d[0];
  size_t hdr = 1, len = 0;
  bool list = b >= 0xc0;
  if (b < 0x80) {
    *out = RlpItem{false, d, 1, 1};
    return true;
  }
  const uint8_t base = list ? 0xc0 : 0x80;
  if (b - base < 56) {
    len = b - base;
  } else {
    const size_t ll = b - base - 55;
    if (ll > 8 || n < 1 + ll) return false;
    for (size_t i = 0; i < ll; i++) len = (len << 8) | d[1 + i];
    hdr = 1 + ll;
    if (len < 56 || d[1] == 0) return false;  // non-canonical length
  }
  if (len > n - hdr) return false;
  if (!list && len == 1 && d[hdr] < 0x80) return false;  // a single small byte must be its own encoding
  *out = RlpItem{list, d + hdr, len, hdr + len};
  return true;
}
bool rlp_children(const RlpItem& list, std::vector<RlpItem>* out) {
  if (!list.is_list) return false;
  size_t off = 0;
  while (off < list.len) {
    RlpItem it;
    if (!rlp_parse(list.payload + off, list.len - off, &it)) return false;
    out->push_back(it);
    off += it.total;
  }
  return true;
}

static bool scalar_ok(const RlpItem& it, size_t max_len) {  // canonical integer: no leading zero byte
  return !it.is_list && it.len <= max_len && (it.len == 0 || it.payload[0] != 0);
}
bool account_decode(const Bytes& rlp, Account* out) {
  RlpItem top;
  if (!rlp_parse(rlp.data(), rlp.size(), &top) || top.total != rlp.size()) return false;
  std::vector<RlpItem> f;
  if (!rlp_children(top, &f) || f.size() != 4) return false;
  if (!scalar_ok(f[0], 32) || !scalar_ok(f[1], 32) || f[2].is_list || f[2].len != 32 || f[3].is_list || f[3].len != 32)
    return false;
  out->nonce_be.assign(f[0].payload, f[0].payload + f[0].len);
  out->balance_be.assign(f[1].payload, f[1].payload + f[1].len);
  std::memcpy(out->storage_root.data(), f[2].payload, 32);
  std::memcpy(out->code_hash.data(), f[3].payload, 32);
  return true;
}
Bytes account_encode(const Account& a) {
  return rlp_list({rlp_scalar_be(a.nonce_be), rlp_scalar_be(a.balance_be), rlp_string(a.storage_root.data(), 32),
                   rlp_string(a.code_hash.data(), 32)});
}
Bytes u256_add_be(const Bytes& a, const Bytes& b, bool* overflow) {
  uint8_t x[32] = {0}, y[32] = {0}, s[32];
  std::memcpy(x + 32 - a.size(), a.data(), a.size());
  std::memcpy(y + 32 - b.size(), b.data(), b.size());
  unsigned carry = 0;
  for (int i = 31; i >= 0; i--) {
    unsigned t = x[i] + y[i] + carry;
    s[i] = (uint8_t)t;
    carry = t >> 8;
  }
  if (overflow) *overflow = carry != 0;
  size_t i = 0;
  while (i < 32 && s[i] == 0) i++;
  return Bytes(s + i, s + 32);
}

// ---------------------------------------------------------------- nodes
Bytes hex_prefix(const Nibbles& k, bool leaf) {
  Bytes out;
  const uint8_t flag = (leaf ? 2 : 0) + (k.size() & 1);
  size_t i = 0;
  if (k.size() & 1) out.push_back((uint8_t)((flag << 4) | k[i++]));
  else out.push_back((uint8_t)(flag << 4));
  for (; i < k.size(); i += 2) out.push_back((uint8_t)((k[i] << 4) | k[i + 1]));
  return out;
}
NodeP make_empty() {
  static const NodeP e = std::make_shared<const Node>();
  return e;
}
NodeP make_hash(const H256& h) {
  auto n = std::make_shared<Node>();
  n->kind = Kind::Hash;
  n->hash = h;
  return n;
}
NodeP make_leaf(Nibbles key, Bytes value) {
  auto n = std::make_shared<Node>();
  n->kind = Kind::Leaf;
  n->key = std::move(key);
  n->value = std::move(value);
  return n;
}
NodeP make_extension(Nibbles key, NodeP child) {
  auto n = std::make_shared<Node>();
  n->kind = Kind::Extension;
  n->key = std::move(key);
  n->child = std::move(child);
  return n;
}
NodeP make_branch(const NodeP (&children)[16], Bytes value) {
  auto n = std::make_shared<Node>();
  n->kind = Kind::Branch;
  for (int i = 0; i < 16; i++) n->children[i] = (children[i] && children[i]->kind != Kind::Empty) ? children[i] : nullptr;
  n->value = std::move(value);
  return n;
}
const char* status_text(Status s) {
  switch (s) {
    case Status::Ok: return "ok";
    case Status::HitHashNode: return "the key's path runs into a hashed-out node";
    case Status::CannotCollapse: return "a delete must collapse a branch into a hashed-out sibling (the witness lacks that node)";
    default: return "malformed trie bytes";
  }
}

static const Bytes& encode(const Node& n);
// child reference inside a parent: raw RLP when shorter than 32 bytes, else keccak as a 32-byte string
static Bytes node_ref(const NodeP& n) {
  if (!n || n->kind == Kind::Empty) return Bytes{0x80};
  if (n->kind == Kind::Hash) return rlp_string(n->hash.data(), 32);
  const Bytes& enc = encode(*n);
  if (enc.size() < 32) return enc;
  H256 h = keccak256(enc);
  return rlp_string(h.data(), 32);
}
static const Bytes& encode(const Node& n) {
  if (n.enc_valid) return n.enc;
  switch (n.kind) {
    case Kind::Branch: {
      std::vector<Bytes> items;
      for (int i = 0; i < 16; i++) items.push_back(node_ref(n.children[i]));
      items.push_back(n.value.empty() ? Bytes{0x80} : rlp_string(n.value));
      n.enc = rlp_list(items);
      break;
    }
    case Kind::Extension: n.enc = rlp_list({rlp_string(hex_prefix(n.key, false)), node_ref(n.child)}); break;
    case Kind::Leaf: n.enc = rlp_list({rlp_string(hex_prefix(n.key, true)), rlp_string(n.value)}); break;
    default: n.enc = Bytes{0x80}; break;
  }
  n.enc_valid = true;
  return n.enc;
}

// The RLP encodings a hasher of the (partial) trie runs Keccak-256 over: every node its parent references by hash (an
// encoding of 32 bytes or more) and the root whatever its length, children before parents; hashed-out subtrees
// contribute nothing.  (What a zkEVM's Keccak table holds for this trie; gi.cpp.)
static void preimages_rec(const NodeP& n, bool is_root, std::vector<Bytes>* out) {
  if (!n || n->kind == Kind::Empty || n->kind == Kind::Hash) return;
  if (n->kind == Kind::Branch)
    for (int i = 0; i < 16; i++) preimages_rec(n->children[i], false, out);
  else if (n->kind == Kind::Extension)
    preimages_rec(n->child, false, out);
  const Bytes& enc = encode(*n);
  if (is_root || enc.size() >= 32) out->push_back(enc);
}
void hashed_node_preimages(const Trie& t, std::vector<Bytes>* out) { preimages_rec(t.root(), true, out); }

Trie::Trie() : root_(make_empty()) {}
Trie Trie::of_hash(const H256& h) { return Trie(make_hash(h)); }

H256 Trie::hash() const {
  if (root_->kind == Kind::Empty) return EMPTY_TRIE_HASH;
  if (root_->kind == Kind::Hash) return root_->hash;
  return keccak256(encode(*root_));
}

static size_t lcp(const Nibbles& a, size_t ao, const Nibbles& b, size_t bo) {
  size_t n = 0;
  while (ao + n < a.size() && bo + n < b.size() && a[ao + n] == b[bo + n]) n++;
  return n;
}
static Nibbles slice(const Nibbles& k, size_t from, size_t to) { return Nibbles(k.begin() + from, k.begin() + to); }
static Nibbles concat(const Nibbles& a, const Nibbles& b) {
  Nibbles r(a);
  r.insert(r.end(), b.begin(), b.end());
  return r;
}
static NodeP with_prefix(const Nibbles& pre, NodeP n) { return pre.empty() ? n : make_extension(pre, std::move(n)); }

// k[o:] is what is left of the key
static Status insert_rec(const NodeP& n, const Nibbles& k, size_t o, const Bytes& v, NodeP* out) {
  switch (n ? n->kind : Kind::Empty) {
    case Kind::Empty:
      *out = make_leaf(slice(k, o, k.size()), v);
      return Status::Ok;
    case Kind::Hash:
      return Status::HitHashNode;
    case Kind::Leaf: {
      const size_t c = lcp(n->key, 0, k, o);
      if (c == n->key.size() && o + c == k.size()) {
        *out = make_leaf(n->key, v);
        return Status::Ok;
      }
      NodeP ch[16];
      Bytes bval;
      if (c == n->key.size()) bval = n->value;  // old key ends at the branch
      else ch[n->key[c]] = make_leaf(slice(n->key, c + 1, n->key.size()), n->value);
      if (o + c == k.size()) bval = v;
      else ch[k[o + c]] = make_leaf(slice(k, o + c + 1, k.size()), v);
      *out = with_prefix(slice(k, o, o + c), make_branch(ch, bval));
      return Status::Ok;
    }
    case Kind::Extension: {
      const size_t c = lcp(n->key, 0, k, o);
      if (c == n->key.size()) {
        NodeP sub;
        Status s = insert_rec(n->child, k, o + c, v, &sub);
        if (s != Status::Ok) return s;
        *out = make_extension(n->key, sub);
        return Status::Ok;
      }
      NodeP ch[16];
      Bytes bval;
      ch[n->key[c]] = with_prefix(slice(n->key, c + 1, n->key.size()), n->child);
      if (o + c == k.size()) bval = v;
      else ch[k[o + c]] = make_leaf(slice(k, o + c + 1, k.size()), v);
      *out = with_prefix(slice(k, o, o + c), make_branch(ch, bval));
      return Status::Ok;
    }
    case Kind::Branch: {
      NodeP ch[16];
      for (int i = 0; i < 16; i++) ch[i] = n->children[i];
      Bytes bval = n->value;
      if (o == k.size()) {
        bval = v;
      } else {
        NodeP sub;
        Status s = insert_rec(n->children[k[o]], k, o + 1, v, &sub);
        if (s != Status::Ok) return s;
        ch[k[o]] = sub;
      }
      *out = make_branch(ch, bval);
      return Status::Ok;
    }
  }
  return Status::Malformed;
}
Status Trie::insert(const Nibbles& k, const Bytes& v) {
  NodeP r;
  Status s = insert_rec(root_, k, 0, v, &r);
  if (s == Status::Ok) root_ = r;
  return s;
}

Status Trie::get(const Nibbles& k, const Bytes** out) const {
  *out = nullptr;
  const Node* n = root_.get();
  size_t o = 0;
  for (;;) {
    switch (n ? n->kind : Kind::Empty) {
      case Kind::Empty: return Status::Ok;
      case Kind::Hash: return Status::HitHashNode;
      case Kind::Leaf:
        if (n->key.size() == k.size() - o && lcp(n->key, 0, k, o) == n->key.size()) *out = &n->value;
        return Status::Ok;
      case Kind::Extension:
        if (lcp(n->key, 0, k, o) != n->key.size()) return Status::Ok;
        o += n->key.size();
        n = n->child.get();
        break;
      case Kind::Branch:
        if (o == k.size()) {
          if (!n->value.empty()) *out = &n->value;
          return Status::Ok;
        }
        n = n->children[k[o++]].get();
        break;
    }
  }
}

// After a removal below it, a node must be brought back to normal form (Yellow Paper: no branch with fewer than
// two entries, no extension over a leaf or an extension).
static Status normalise_branch(const NodeP (&ch)[16], const Bytes& bval, NodeP* out) {
  int cnt = 0, last = -1;
  for (int i = 0; i < 16; i++)
    if (ch[i] && ch[i]->kind != Kind::Empty) { cnt++; last = i; }
  if (cnt >= 2 || (cnt == 1 && !bval.empty())) {
    *out = make_branch(ch, bval);
    return Status::Ok;
  }
  if (cnt == 0) {
    *out = bval.empty() ? make_empty() : make_leaf(Nibbles{}, bval);
    return Status::Ok;
  }
  const NodeP& c = ch[last];
  const Nibbles pre{(uint8_t)last};
  switch (c->kind) {
    case Kind::Leaf: *out = make_leaf(concat(pre, c->key), c->value); return Status::Ok;
    case Kind::Extension: *out = make_extension(concat(pre, c->key), c->child); return Status::Ok;
    case Kind::Branch: *out = make_extension(pre, c); return Status::Ok;
    default: return Status::CannotCollapse;  // a hash node: its kind is unknown, the merged node cannot be formed
  }
}
static Status remove_rec(const NodeP& n, const Nibbles& k, size_t o, bool* existed, NodeP* out) {
  *out = n;
  switch (n ? n->kind : Kind::Empty) {
    case Kind::Empty: return Status::Ok;
    case Kind::Hash: return Status::HitHashNode;
    case Kind::Leaf:
      if (n->key.size() == k.size() - o && lcp(n->key, 0, k, o) == n->key.size()) {
        *existed = true;
        *out = make_empty();
      }
      return Status::Ok;
    case Kind::Extension: {
      if (lcp(n->key, 0, k, o) != n->key.size()) return Status::Ok;
      NodeP sub;
      Status s = remove_rec(n->child, k, o + n->key.size(), existed, &sub);
      if (s != Status::Ok || !*existed) return s;
      switch (sub->kind) {
        case Kind::Empty: *out = make_empty(); break;
        case Kind::Leaf: *out = make_leaf(concat(n->key, sub->key), sub->value); break;
        case Kind::Extension: *out = make_extension(concat(n->key, sub->key), sub->child); break;
        default: *out = make_extension(n->key, sub); break;
      }
      return Status::Ok;
    }
    case Kind::Branch: {
      NodeP ch[16];
      for (int i = 0; i < 16; i++) ch[i] = n->children[i];
      Bytes bval = n->value;
      if (o == k.size()) {
        if (bval.empty()) return Status::Ok;
        *existed = true;
        bval.clear();
      } else {
        NodeP sub;
        Status s = remove_rec(n->children[k[o]], k, o + 1, existed, &sub);
        if (s != Status::Ok || !*existed) return s;
        ch[k[o]] = sub;
      }
      return normalise_branch(ch, bval, out);
    }
  }
  return Status::Malformed;
}
Status Trie::remove(const Nibbles& k, bool* existed) {
  bool ex = false;
  NodeP r;
  Status s = remove_rec(root_, k, 0, &ex, &r);
  if (existed) *existed = ex;
  if (s == Status::Ok && ex) root_ = r;
  return s;
}

static void items_rec(const NodeP& n, Nibbles& path, std::vector<Item>* out) {
  if (!n) return;
  switch (n->kind) {
    case Kind::Empty: return;
    case Kind::Hash: out->push_back(Item{path, true, Bytes(n->hash.begin(), n->hash.end())}); return;
    case Kind::Leaf: out->push_back(Item{concat(path, n->key), false, n->value}); return;
    case Kind::Extension: {
      const size_t keep = path.size();
      path.insert(path.end(), n->key.begin(), n->key.end());
      items_rec(n->child, path, out);
      path.resize(keep);
      return;
    }
    case Kind::Branch:
      for (int i = 0; i < 16; i++) {
        path.push_back((uint8_t)i);
        items_rec(n->children[i], path, out);
        path.pop_back();
      }
      if (!n->value.empty()) out->push_back(Item{path, false, n->value});
      return;
  }
}
void Trie::items(std::vector<Item>* out) const {
  Nibbles path;
  items_rec(root_, path, out);
}

static NodeP hashed_out(const NodeP& n) {
  if (!n || n->kind == Kind::Empty) return make_empty();
  if (n->kind == Kind::Hash) return n;
  const Bytes& enc = encode(*n);
  if (enc.size() < 32) return n;  // embedded in its parent: not replaceable by a hash
  return make_hash(keccak256(enc));
}
// keys: (key, offset) pairs still alive at this node
static Status subset_rec(const NodeP& n, const std::vector<std::pair<const Nibbles*, size_t>>& keys, NodeP* out) {
  if (keys.empty()) {
    *out = hashed_out(n);
    return Status::Ok;
  }
  switch (n ? n->kind : Kind::Empty) {
    case Kind::Empty: *out = make_empty(); return Status::Ok;
    case Kind::Hash: return Status::HitHashNode;
    case Kind::Leaf: *out = n; return Status::Ok;  // proves presence or absence of every key that got here
    case Kind::Extension: {
      std::vector<std::pair<const Nibbles*, size_t>> down;
      for (auto& ko : keys)
        if (lcp(n->key, 0, *ko.first, ko.second) == n->key.size()) down.push_back({ko.first, ko.second + n->key.size()});
      NodeP sub;
      Status s = subset_rec(n->child, down, &sub);
      if (s != Status::Ok) return s;
      *out = make_extension(n->key, sub);
      return Status::Ok;
    }
    case Kind::Branch: {
      NodeP ch[16];
      for (int i = 0; i < 16; i++) {
        std::vector<std::pair<const Nibbles*, size_t>> down;
        for (auto& ko : keys)
          if (ko.second < ko.first->size() && (*ko.first)[ko.second] == i) down.push_back({ko.first, ko.second + 1});
        Status s = subset_rec(n->children[i], down, &ch[i]);
        if (s != Status::Ok) return s;
      }
      *out = make_branch(ch, n->value);
      return Status::Ok;
    }
  }
  return Status::Malformed;
}
Status Trie::subset(const std::vector<Nibbles>& keys, Trie* out) const {
  std::vector<std::pair<const Nibbles*, size_t>> ks;
  for (auto& k : keys) ks.push_back({&k, 0});
  NodeP r;
  Status s = subset_rec(root_, ks, &r);
  if (s == Status::Ok) *out = Trie(r);
  return s;
}
Trie Trie::fully_hashed() const {
  if (root_->kind == Kind::Empty) return Trie();
  return Trie::of_hash(hash());
}

// ---------------------------------------------------------------- byte form
// node := 0x00                                            empty
//       | 0x01 hash[32]                                   hashed-out subtree
//       | 0x02 mask:u16le vlen:u32le value node*          branch: present children in nibble order
//       | 0x03 nkey:u8 nibble[nkey] node                  extension
//       | 0x04 nkey:u8 nibble[nkey] vlen:u32le value      leaf
static void put_u32(Bytes* out, uint32_t v) {
  for (int i = 0; i < 4; i++) out->push_back((uint8_t)(v >> (8 * i)));
}
static void ser_rec(const NodeP& n, Bytes* out) {
  switch (n ? n->kind : Kind::Empty) {
    case Kind::Empty: out->push_back(0); return;
    case Kind::Hash:
      out->push_back(1);
      out->insert(out->end(), n->hash.begin(), n->hash.end());
      return;
    case Kind::Branch: {
      uint16_t mask = 0;
      for (int i = 0; i < 16; i++)
        if (n->children[i]) mask |= (uint16_t)(1u << i);
      out->push_back(2);
      out->push_back((uint8_t)mask);
      out->push_back((uint8_t)(mask >> 8));
      put_u32(out, (uint32_t)n->value.size());
      out->insert(out->end(), n->value.begin(), n->value.end());
      for (int i = 0; i < 16; i++)
        if (n->children[i]) ser_rec(n->children[i], out);
      return;
    }
    case Kind::Extension:
      out->push_back(3);
      out->push_back((uint8_t)n->key.size());
      out->insert(out->end(), n->key.begin(), n->key.end());
      ser_rec(n->child, out);
      return;
    case Kind::Leaf:
      out->push_back(4);
      out->push_back((uint8_t)n->key.size());
      out->insert(out->end(), n->key.begin(), n->key.end());
      put_u32(out, (uint32_t)n->value.size());
      out->insert(out->end(), n->value.begin(), n->value.end());
      return;
  }
}
void Trie::serialize(Bytes* out) const { ser_rec(root_, out); }

static bool de_rec(const uint8_t* d, size_t n, size_t* pos, int depth, NodeP* out) {
  if (depth > 200 || *pos >= n) return false;
  const uint8_t tag = d[(*pos)++];
  auto need = [&](size_t k) { return n - *pos >= k; };
  auto u32 = [&](uint32_t* v) {
    if (!need(4)) return false;
    *v = (uint32_t)d[*pos] | ((uint32_t)d[*pos + 1] << 8) | ((uint32_t)d[*pos + 2] << 16) | ((uint32_t)d[*pos + 3] << 24);
    *pos += 4;
    return true;
  };
  auto key = [&](Nibbles* k) {
    if (!need(1)) return false;
    const size_t nk = d[(*pos)++];
    if (!need(nk)) return false;
    k->assign(d + *pos, d + *pos + nk);
    *pos += nk;
    for (uint8_t x : *k)
      if (x > 15) return false;
    return true;
  };
  switch (tag) {
    case 0: *out = make_empty(); return true;
    case 1: {
      if (!need(32)) return false;
      H256 h;
      std::memcpy(h.data(), d + *pos, 32);
      *pos += 32;
      *out = make_hash(h);
      return true;
    }
    case 2: {
      if (!need(2)) return false;
      const uint16_t mask = (uint16_t)(d[*pos] | (d[*pos + 1] << 8));
      *pos += 2;
      uint32_t vl;
      if (!u32(&vl) || !need(vl)) return false;
      Bytes val(d + *pos, d + *pos + vl);
      *pos += vl;
      NodeP ch[16];
      for (int i = 0; i < 16; i++)
        if (mask & (1u << i)) {
          if (!de_rec(d, n, pos, depth + 1, &ch[i])) return false;
          if (ch[i]->kind == Kind::Empty) return false;
        }
      *out = make_branch(ch, val);
      return true;
    }
    case 3: {
      Nibbles k;
      NodeP c;
      if (!key(&k) || k.empty() || !de_rec(d, n, pos, depth + 1, &c)) return false;
      *out = make_extension(k, c);
      return true;
    }
    case 4: {
      Nibbles k;
      uint32_t vl;
      if (!key(&k) || !u32(&vl) || !need(vl)) return false;
      *out = make_leaf(k, Bytes(d + *pos, d + *pos + vl));
      *pos += vl;
      return true;
    }
    default: return false;
  }
}
Status Trie::deserialize(const uint8_t* d, size_t n, size_t* used, Trie* out) {
  size_t pos = 0;
  NodeP r;
  if (!de_rec(d, n, &pos, 0, &r)) return Status::Malformed;
  if (used) *used = pos;
  *out = Trie(r);
  return Status::Ok;
}

}  // namespace mpt
