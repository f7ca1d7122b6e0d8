// api_entry_container.inc.h — C ABI entry point: the container framing of a dumped index file (inside extern "C")
// Part of zvec_hip_api.hip (one translation unit; included in order, not standalone).  Host-only code: no HIP call.
//
// IndexUnpacker::unpack (src/include/zvec/core/framework/index_unpacker.h:103-330) over a whole file image laid out as
// IndexFormat says (index_format.h:26-95): [MetaHeader 64 B][content: segments' data + padding][segment metas +
// NUL-terminated ids][MetaFooter 128 B], possibly chained through MetaFooter::next_meta_header_offset.  The plugin
// itself receives segments from zvec's IndexStorage and never needs this; it lets a caller that holds only the FILE
// (tools, tests, a loader outside zvec) find the "flat.*" / "ivf.*" payloads the segment loaders take.

}  // extern "C"

namespace {
struct RefMetaHeader {          // IndexFormat::MetaHeader
  uint32_t header_crc; uint16_t reserved1_, version; uint32_t revision, magic; uint16_t meta_header_size, meta_footer_size;
  uint32_t meta_footer_offset; uint64_t content_offset, setup_time, reserved3_[3];
};
struct RefMetaFooter {          // IndexFormat::MetaFooter
  uint32_t footer_crc, segments_meta_crc, content_crc, segment_count, segments_meta_size, reserved1_;
  uint64_t content_size, content_padding_size, check_point, update_time, reserved2_[7], next_meta_header_offset, total_size;
};
struct RefSegmentMeta {         // IndexFormat::SegmentMeta
  uint32_t segment_id_offset, data_crc; uint64_t data_index, data_size, padding_size;
};
static_assert(sizeof(RefMetaHeader) == 64 && sizeof(RefMetaFooter) == 128 && sizeof(RefSegmentMeta) == 32, "index_format.h layouts");
}  // namespace

extern "C" {

int zvec_hip_container_segments(const void *image, uint64_t bytes, int checksum, zvec_hip_segment_t *out, uint32_t cap,
                                uint32_t *count) {
  if (!image || !count || (cap && !out)) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
  const uint8_t *b = static_cast<const uint8_t *>(image);
  uint32_t n = 0;
  uint64_t h0 = 0;                                   // current_header_start_offset_
  for (uint32_t hop = 0; hop < 1024; ++hop) {        // (a chain cannot be longer than the file has headers)
    RefMetaHeader hd;
    if (h0 > bytes || bytes - h0 < sizeof(hd)) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
    memcpy(&hd, b + h0, sizeof(hd));
    if (hd.meta_header_size != sizeof(hd) || hd.meta_footer_size != sizeof(RefMetaFooter)) return ZVEC_HIP_ERR_MISMATCH;
    // the crc field comes first and was zero when the crc was taken: feeding it back as the initial value cancels it
    if (crc32c_update(&hd, sizeof(hd), hd.header_crc) != hd.header_crc) return ZVEC_HIP_ERR_MISMATCH;
    const uint64_t total = bytes;                    // unpack(read_data, total, ...) passes the file size
    const uint64_t foff = ((int32_t)hd.meta_footer_offset < 0) ? total + (int64_t)(int32_t)hd.meta_footer_offset : hd.meta_footer_offset;
    if (foff > total || total - foff < sizeof(RefMetaFooter) || h0 + foff + sizeof(RefMetaFooter) > bytes) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
    RefMetaFooter ft;
    memcpy(&ft, b + h0 + foff, sizeof(ft));
    if (ft.content_size + ft.content_padding_size + hd.content_offset > ft.total_size) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
    if (crc32c_update(&ft, sizeof(ft), ft.footer_crc) != ft.footer_crc) return ZVEC_HIP_ERR_MISMATCH;
    if ((uint64_t)sizeof(RefSegmentMeta) * ft.segment_count > ft.segments_meta_size) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
    if (foff < ft.segments_meta_size) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
    const uint64_t moff = foff - ft.segments_meta_size;
    const uint8_t *meta = b + h0 + moff;
    if (crc32c_update(meta, ft.segments_meta_size, 0u) != ft.segments_meta_crc) return ZVEC_HIP_ERR_MISMATCH;
    if (checksum && ft.content_crc != 0) {
      if (h0 + sizeof(hd) + ft.content_size > bytes) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
      if (crc32c_update(b + h0 + sizeof(hd), ft.content_size, 0u) != ft.content_crc) return ZVEC_HIP_ERR_MISMATCH;
    }
    for (uint32_t i = 0; i < ft.segment_count; ++i) {
      RefSegmentMeta sm;
      memcpy(&sm, meta + (size_t)i * sizeof(sm), sizeof(sm));
      if (sm.segment_id_offset >= ft.segments_meta_size || sm.data_index > ft.content_size ||
          sm.data_size > ft.content_size - sm.data_index)
        return ZVEC_HIP_ERR_INVALID_ARGUMENT;
      const char *id = reinterpret_cast<const char *>(meta) + sm.segment_id_offset;
      const size_t idmax = ft.segments_meta_size - sm.segment_id_offset;
      const size_t idlen = strnlen(id, idmax);
      if (idlen == idmax) return ZVEC_HIP_ERR_INVALID_ARGUMENT;             // no terminator inside the meta block
      const uint64_t off = sm.data_index + hd.content_offset + h0;
      if (off > bytes || sm.data_size > bytes - off) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
      if (n < cap) {
        zvec_hip_segment_t &o = out[n];
        memset(&o, 0, sizeof(o));
        memcpy(o.id, id, std::min(idlen, sizeof(o.id) - 1));
        o.offset = off; o.size = sm.data_size; o.padding = sm.padding_size; o.crc = sm.data_crc;
      }
      ++n;
    }
    if (ft.next_meta_header_offset == 0) break;
    h0 = ft.next_meta_header_offset;
  }
  *count = n;
  return n > cap && cap ? ZVEC_HIP_ERR_OUT_OF_RANGE : 0;
}
