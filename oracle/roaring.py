"""CPU restatement of the composite document filter and of the roaring "portable" serialisation it is fed with.

TEST INFRASTRUCTURE ONLY (see oracle/zvec_oracle.h): tests build bitmaps with this module, hand the bytes to
the product (zvec_hip_*_build_filter) and compare the exclude bits with `doc_filter_mask` below.

What the reference does (and what `doc_filter_mask` restates):
  DocFilter::is_filtered(id)            src/db/sqlengine/planner/doc_filter.cc:74-87
      deleted(id) || !invert.contains(id) || !forward[id]          (each term optional; forward out of range => kept)
  DeleteStore::Filter                   src/db/index/common/delete_store.h:61-72     -> bitmap_.contains(id)
  ConcurrentRoaringBitmap64::contains   src/db/common/concurrent_roaring_bitmap.h:196-203  (32-bit bitmap: (uint32_t)id)
  InvertedSearchResult::Filter/contains .../inverted_search_result.h:34-50,70-76     (uint32_t id)
  file image of a delete store          concurrent_roaring_bitmap.h:186-192 (BitmapMetaHeader, 64 B) + .cc:52-125

PARITY UNPINNED for the byte format: the bitmap container library is a third-party dependency that is absent from
/root/reference (thirdparty/CRoaring is an empty submodule directory, pinned at CRoaring 2.0.4) and the reference
holds no serialised bitmap fixtures.  The writer below follows the published RoaringFormatSpec ("portable" format:
cookies 12346 / 12347, descriptive header, offset header, array / bitset / run containers) that
roaring_bitmap_portable_serialize implements; the semantics (which ids are excluded) are pinned by the call sites above.
"""
import struct

import numpy as np

SERIAL_COOKIE_NO_RUN = 12346
SERIAL_COOKIE = 12347
NO_OFFSET_THRESHOLD = 4
FILE_MAGIC = 0x362DDA444AC1B99A


def _runs_of(vals):
    """sorted unique u16 values -> list of (start, length-1)"""
    if len(vals) == 0:
        return []
    v = np.asarray(vals, np.int64)
    brk = np.nonzero(np.diff(v) != 1)[0]
    starts = np.concatenate(([v[0]], v[brk + 1]))
    ends = np.concatenate((v[brk], [v[-1]]))
    return list(zip(starts.tolist(), (ends - starts).tolist()))


def serialize32(ids, run_optimize=False):
    """portable serialisation of a set of uint32 ids.  run_optimize=True picks a run container wherever it is
    smaller than the array / bitset form (what roaring_bitmap_run_optimize does before serialising)."""
    ids = np.unique(np.asarray(ids, np.uint64))
    assert ids.size == 0 or int(ids[-1]) < (1 << 32)
    ids = ids.astype(np.uint32)
    keys = (ids >> 16).astype(np.uint32)
    conts = []
    for k in np.unique(keys):
        low = (ids[keys == k] & 0xFFFF).astype(np.uint16)
        card = int(low.size)
        runs = _runs_of(low)
        plain_size = 8192 if card > 4096 else 2 * card
        is_run = run_optimize and (2 + 4 * len(runs)) < plain_size
        if is_run:
            payload = struct.pack("<H", len(runs)) + b"".join(struct.pack("<HH", s, l) for s, l in runs)
        elif card > 4096:
            bits = np.zeros(65536, np.uint8)
            bits[low.astype(np.int64)] = 1
            payload = np.packbits(bits, bitorder="little").tobytes()
        else:
            payload = low.astype("<u2").tobytes()
        conts.append((int(k), card, is_run, payload))
    n = len(conts)
    has_run = any(c[2] for c in conts)
    out = bytearray()
    if has_run:
        out += struct.pack("<I", SERIAL_COOKIE | ((n - 1) << 16))
        flags = np.zeros((n + 7) // 8 * 8, np.uint8)
        for i, c in enumerate(conts):
            flags[i] = 1 if c[2] else 0
        out += np.packbits(flags, bitorder="little").tobytes()[: (n + 7) // 8]
    else:
        out += struct.pack("<II", SERIAL_COOKIE_NO_RUN, n)
    for k, card, _, _ in conts:
        out += struct.pack("<HH", k, card - 1)
    if (not has_run) or n >= NO_OFFSET_THRESHOLD:
        off = len(out) + 4 * n
        for c in conts:
            out += struct.pack("<I", off)
            off += len(c[3])
    for c in conts:
        out += c[3]
    return bytes(out)


def serialize64map(ids, run_optimize=False):
    """roaring::Roaring64Map::write(portable): u64 #buckets, then per bucket u32 high + portable 32-bit stream"""
    ids = np.unique(np.asarray(ids, np.uint64))
    highs = (ids >> np.uint64(32)).astype(np.uint64)
    out = bytearray()
    uh = np.unique(highs)
    out += struct.pack("<Q", len(uh))
    for h in uh:
        out += struct.pack("<I", int(h))
        out += serialize32(ids[highs == h] & np.uint64(0xFFFFFFFF), run_optimize)
    return bytes(out)


def crc32c(data, crc=0):
    """ailego::Crc32c::Hash (src/ailego/hash/crc32c.cc:626-634): raw CRC-32C update, init `crc`, no inversions"""
    table = getattr(crc32c, "_t", None)
    if table is None:
        table = []
        for i in range(256):
            c = i
            for _ in range(8):
                c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
            table.append(c)
        crc32c._t = table
    for b in bytes(data):
        crc = table[(crc ^ b) & 0xFF] ^ (crc >> 8)
    return crc


def file_image(payload, is_32bit, timestamp=0):
    """a delete-store file: BitmapMetaHeader {u64 magic, u32 is_32bit, u32 crc32c(payload), u64 time, u32[10]} + payload"""
    return struct.pack("<QIIQ", FILE_MAGIC, 1 if is_32bit else 0, crc32c(payload), timestamp) + b"\0" * 40 + payload


def doc_filter_mask(keys, deleted=None, deleted_is32=True, invert=None, forward=None):
    """exclude mask (bool per position) of the composite filter for positions holding `keys`"""
    keys = np.asarray(keys, np.uint64)
    ex = np.zeros(keys.size, bool)
    if deleted is not None:
        probe = (keys & np.uint64(0xFFFFFFFF)) if deleted_is32 else keys
        ex |= np.isin(probe, np.asarray(deleted, np.uint64))
    if invert is not None:
        ex |= ~np.isin(keys & np.uint64(0xFFFFFFFF), np.asarray(invert, np.uint64))
    if forward is not None:
        f = np.asarray(forward, bool)
        inr = keys < np.uint64(f.size)
        idx = np.where(inr, keys, 0).astype(np.int64)
        ex |= inr & ~f[idx]
    return ex


def mask_to_words(mask):
    m = np.asarray(mask, bool)
    pad = (-m.size) % 64
    bits = np.concatenate([m, np.zeros(pad, bool)])
    return np.packbits(bits, bitorder="little").view(np.uint64)
