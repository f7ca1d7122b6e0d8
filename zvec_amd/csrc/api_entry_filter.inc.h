// api_entry_filter.inc.h — C ABI entry points: predicate materialisation, query reformers, crc32c (inside extern "C")
// Part of zvec_hip_api.hip (one translation unit; included in order, not standalone).

// ---- predicate materialisation ------------------------------------------------------------------------------
namespace {

uint32_t crc32c_update(const void *data, uint64_t len, uint32_t crc) {
  static uint32_t table[256];
  static std::once_flag once;
  std::call_once(once, [] {
    for (uint32_t i = 0; i < 256; ++i) {
      uint32_t c = i;
      for (int k = 0; k < 8; ++k) c = (c & 1u) ? (c >> 1) ^ 0x82F63B78u : (c >> 1);
      table[i] = c;
    }
  });
  const uint8_t *p = static_cast<const uint8_t *>(data);
  for (uint64_t i = 0; i < len; ++i) crc = table[(crc ^ p[i]) & 0xffu] ^ (crc >> 8);
  return crc;
}

// container directory of one (or, for a 64-bit map, several) portable 32-bit roaring streams
struct RoaringDir {
  std::vector<uint64_t> ckey, coff;
  std::vector<uint32_t> cinfo;
};

inline uint32_t rd_u16(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8); }
inline uint32_t rd_u32(const uint8_t *p) { return rd_u16(p) | (rd_u16(p + 2) << 16); }
inline uint64_t rd_u64(const uint8_t *p) { return (uint64_t)rd_u32(p) | ((uint64_t)rd_u32(p + 4) << 32); }

// RoaringFormatSpec "portable" layout (what roaring_bitmap_portable_serialize of CRoaring 2.0.4 writes):
//   cookie  : u32 12346 + u32 container count               (no run containers)
//           | u16 12347, u16 count-1, ceil(count/8) bytes of run flags
//   header  : count x (u16 key, u16 cardinality-1)
//   offsets : count x u32, present unless (run cookie && count < 4)
//   payload : per container — run: u16 n_runs + n_runs x (u16 start, u16 length-1);
//             cardinality > 4096: 1024 x u64 bitset; else cardinality x u16 sorted values
// Returns the bytes consumed, or 0 for a malformed stream.  `base` = offset of b[0] in the uploaded buffer.
uint64_t parse_roaring32(const uint8_t *b, uint64_t len, uint64_t high, uint64_t base, RoaringDir &dir) {
  if (len < 4) return 0;
  const uint32_t cookie = rd_u32(b);
  uint64_t pos;
  uint32_t n;
  const uint8_t *runflags = nullptr;
  if ((cookie & 0xffffu) == 12347u) {
    n = (cookie >> 16) + 1;
    runflags = b + 4;
    pos = 4 + (n + 7) / 8;
  } else if (cookie == 12346u) {
    if (len < 8) return 0;
    n = rd_u32(b + 4);
    pos = 8;
  } else {
    return 0;
  }
  if (n > 65536u || pos + (uint64_t)4 * n > len) return 0;
  const uint8_t *desc = b + pos;
  pos += (uint64_t)4 * n;
  if (runflags == nullptr || n >= 4) {
    if (pos + (uint64_t)4 * n > len) return 0;
    pos += (uint64_t)4 * n;
  }
  uint32_t prev_key = 0;
  for (uint32_t i = 0; i < n; ++i) {
    const uint32_t key = rd_u16(desc + 4 * i), card = rd_u16(desc + 4 * i + 2) + 1;
    if (i > 0 && key <= prev_key) return 0;          // keys strictly ascending
    prev_key = key;
    const bool is_run = runflags && ((runflags[i >> 3] >> (i & 7)) & 1u);
    uint32_t type, cnt;
    uint64_t size, payload = pos;
    if (is_run) {
      if (pos + 2 > len) return 0;
      cnt = rd_u16(b + pos);
      type = 2; payload = pos + 2; size = 2 + (uint64_t)4 * cnt;
    } else if (card > 4096u) {
      type = 1; cnt = card; size = 8192;
    } else {
      type = 0; cnt = card; size = (uint64_t)2 * card;
    }
    if (pos + size > len) return 0;
    dir.ckey.push_back((high << 16) | key);
    dir.cinfo.push_back(type | (cnt << 2));
    dir.coff.push_back(base + payload);
    pos += size;
  }
  return pos;
}

// roaring::Roaring64Map::write(portable): u64 map size, then per entry u32 high key + a portable 32-bit stream
bool parse_roaring64map(const uint8_t *b, uint64_t len, uint64_t base, RoaringDir &dir) {
  if (len < 8) return false;
  const uint64_t m = rd_u64(b);
  uint64_t pos = 8;
  uint64_t prev = 0;
  for (uint64_t i = 0; i < m; ++i) {
    if (pos + 4 > len) return false;
    const uint64_t high = rd_u32(b + pos);
    if (i > 0 && high <= prev) return false;
    prev = high;
    pos += 4;
    const uint64_t used = parse_roaring32(b + pos, len - pos, high, base + pos, dir);
    if (used == 0) return false;
    pos += used;
  }
  return true;
}

struct BitmapFileHeader {     // concurrent_roaring_bitmap.h:186-192
  uint64_t magic;
  uint32_t is_32bit;
  uint32_t checksum;
  uint64_t timestamp;
  uint32_t reserved_[10];
};
static_assert(sizeof(BitmapFileHeader) == 64, "BitmapMetaHeader is 64 bytes");
constexpr uint64_t ROARING_FILE_MAGIC = 0x362DDA444AC1B99Aull;

struct DeviceRoaring {
  Scoped<uint8_t> bytes;
  Scoped<uint64_t> ckey, coff;
  Scoped<uint32_t> cinfo;
  RoaringView view{};
};

int upload_roaring(const void *data, uint64_t len, int kind, DeviceRoaring &out, hipStream_t s) {
  out.view = RoaringView{};
  if (kind == ZVEC_HIP_ROARING_NONE || data == nullptr) return 0;
  const uint8_t *b = static_cast<const uint8_t *>(data);
  uint64_t off = 0;
  if (kind == ZVEC_HIP_ROARING_FILE) {
    if (len < sizeof(BitmapFileHeader)) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
    BitmapFileHeader hd;
    memcpy(&hd, b, sizeof(hd));
    if (hd.magic != ROARING_FILE_MAGIC) return ZVEC_HIP_ERR_MISMATCH;
    off = sizeof(hd);
    if (crc32c_update(b + off, len - off, 0u) != hd.checksum) return ZVEC_HIP_ERR_MISMATCH;
    kind = hd.is_32bit ? ZVEC_HIP_ROARING_32 : ZVEC_HIP_ROARING_64MAP;
  }
  RoaringDir dir;
  if (kind == ZVEC_HIP_ROARING_32) {
    if (parse_roaring32(b + off, len - off, 0, off, dir) == 0) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
    out.view.trunc32 = 1;
  } else if (kind == ZVEC_HIP_ROARING_64MAP) {
    if (!parse_roaring64map(b + off, len - off, off, dir)) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
  } else {
    return ZVEC_HIP_ERR_INVALID_ARGUMENT;
  }
  const size_t nc = dir.ckey.size();
  out.view.present = 1;
  out.view.nc = (uint32_t)nc;
  ZRET(out.bytes.alloc(std::max<uint64_t>(len, 1)));
  ZCHK(hipMemcpyAsync(out.bytes, b, len, hipMemcpyHostToDevice, s));
  if (nc) {
    ZRET(out.ckey.alloc(nc));
    ZRET(out.coff.alloc(nc));
    ZRET(out.cinfo.alloc(nc));
    ZCHK(hipMemcpyAsync(out.ckey, dir.ckey.data(), nc * 8, hipMemcpyHostToDevice, s));
    ZCHK(hipMemcpyAsync(out.coff, dir.coff.data(), nc * 8, hipMemcpyHostToDevice, s));
    ZCHK(hipMemcpyAsync(out.cinfo, dir.cinfo.data(), nc * 4, hipMemcpyHostToDevice, s));
  }
  ZCHK(hipStreamSynchronize(s));     // `dir` (pageable host memory) may go away now
  out.view.ckey = out.ckey; out.view.coff = out.coff; out.view.cinfo = out.cinfo; out.view.bytes = out.bytes;
  return 0;
}

int build_filter(zvec_hip_ctx_s *c, int device, const uint64_t *d_keys, uint64_t n, const uint64_t *d_dense0,
                 const uint32_t *d_tile0, uint32_t nlist, const zvec_hip_doc_filter_t *f, uint64_t *out_words,
                 int out_on_device, void *stream) {
  if (!c || !f || !out_words) return ZVEC_HIP_ERR_INVALID_ARGUMENT;   // (the caller holds c->mu)
  ZCHK(hipSetDevice(device));
  hipStream_t s = stream ? reinterpret_cast<hipStream_t>(stream) : c->cur;
  const uint64_t words = (n + 63) / 64;
  if (words == 0) return 0;
  DeviceRoaring del, inv;
  ZRET(upload_roaring(f->delete_bitmap, f->delete_bytes, f->delete_kind, del, s));
  ZRET(upload_roaring(f->invert_bitmap, f->invert_bytes, f->invert_bitmap ? ZVEC_HIP_ROARING_32 : ZVEC_HIP_ROARING_NONE, inv, s));
  Scoped<uint8_t> fwd;
  if (f->forward_bits) {
    const uint64_t fb = (f->forward_len + 7) / 8;
    ZRET(fwd.alloc(std::max<uint64_t>(fb, 1)));
    ZCHK(hipMemcpyAsync(fwd, f->forward_bits, fb, hipMemcpyHostToDevice, s));
  }
  Scoped<uint64_t> tmp;
  uint64_t *d_out = out_words;
  if (!out_on_device) {
    ZRET(tmp.alloc(words));
    d_out = tmp;
  }
  DocFilterArgs a{};
  a.keys = d_keys; a.n = n; a.list_dense0 = d_dense0; a.list_tile0 = d_tile0; a.nlist = nlist;
  a.del = del.view; a.inv = inv.view;
  a.forward = f->forward_bits ? static_cast<const uint8_t *>(fwd) : nullptr; a.forward_len = f->forward_len;
  a.out = d_out;
  hipLaunchKernelGGL(doc_filter_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, a);
  ZCHK(hipGetLastError());
  if (!out_on_device) ZCHK(hipMemcpyAsync(out_words, d_out, words * 8, hipMemcpyDeviceToHost, s));
  ZCHK(hipStreamSynchronize(s));     // the temporaries above are freed on return
  return 0;
}

}  // namespace

extern "C" int zvec_hip_reform_queries_dev(zvec_hip_ctx_t ctx, const float *d_in, uint32_t count, uint32_t dim, int cosine,
                                           int out_dtype, void *d_out, void *stream) {
  if (!ctx || !d_in || !d_out || dim == 0) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
  if (out_dtype != ZVEC_HIP_DT_FP32 && out_dtype != ZVEC_HIP_DT_FP16) return ZVEC_HIP_ERR_UNSUPPORTED;
  if (count == 0) return 0;
  std::lock_guard<std::mutex> g(ctx->mu);
  ZCHK(hipSetDevice(ctx->device));
  hipStream_t s = stream ? reinterpret_cast<hipStream_t>(stream) : ctx->cur;
  hipLaunchKernelGGL(reform_queries_kernel, dim3((count + 15) / 16), dim3(256), 0, s, d_in, count, dim, cosine ? 1 : 0,
                     out_dtype == ZVEC_HIP_DT_FP16 ? 1 : 0, d_out);
  ZCHK(hipGetLastError());
  return 0;
}

extern "C" uint32_t zvec_hip_crc32c(const void *data, uint64_t len, uint32_t crc) {
  return (data || len == 0) ? crc32c_update(data, len, crc) : crc;
}

extern "C" int zvec_hip_flat_build_filter(zvec_hip_flat_t h, zvec_hip_ctx_t ctx, const zvec_hip_doc_filter_t *filter,
                                          uint64_t *out_words, int out_on_device, void *stream) {
  if (!h) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
  zvec_hip_ctx_s *c = ctx ? ctx : h->defctx;
  std::lock_guard<std::mutex> g(c->mu);              // lock order everywhere: context, then the store's rw lock
  std::shared_lock<FairSharedMutex> r(h->rw);
  if (h->append_pending) {
    ZCHK(hipSetDevice(h->device));
    ZRET(flat_wait_appends(h, stream ? reinterpret_cast<hipStream_t>(stream) : c->cur));
  }
  return build_filter(c, h->device, h->st.keys, h->st.n, nullptr, nullptr, 0, filter, out_words,
                      out_on_device, stream);
}

extern "C" int zvec_hip_ivf_build_filter(zvec_hip_ivf_t h, zvec_hip_ctx_t ctx, const zvec_hip_doc_filter_t *filter,
                                         uint64_t *out_words, int out_on_device, void *stream) {
  if (!h) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
  if (!h->loaded) return ZVEC_HIP_ERR_NO_INDEX_LOADED;
  zvec_hip_ctx_s *c = ctx ? ctx : h->defctx;
  std::lock_guard<std::mutex> g(c->mu);
  return build_filter(c, h->device, h->lists.keys, h->count_local, h->d_dense0, h->d_tile0, h->nlist,
                      filter, out_words, out_on_device, stream);
}
