"""CPU checks of the predicate-materialisation helpers: the CRC the delete-store file header carries against the
reference's own known answers (tests/ailego/hash/crc32c_test.cc:22-50), through both the oracle restatement and the
C ABI (host-only code, no GPU), and the structure of the roaring writer the GPU tests feed the product with."""
import ctypes as C
import struct

import numpy as np

from oracle import roaring as R

# tests/ailego/hash/crc32c_test.cc:24-49
CRC_KNOWN = [(b"", 0, 0x0), (b"123456789", 0, 0x58E3FA20), (b"whiz bang boom", 0, 0x8CAE40C8),
             (b"whiz bang boom", 5678, 0xDF19F0C8), (b"foo bar baz", 0, 0xF58C78AC), (b"foo bar baz", 1234, 0x348DACCE)]
CRC_PREFIXES = [3263744690, 2184491954, 1881115848, 3193814825, 1570985216, 371133708, 2843540871, 3970904592,
                1491335712, 551906596]


def test_crc32c_known_answers_oracle_and_library():
    from zvec_amd import _lib
    L = _lib.lib()
    for data, init, want in CRC_KNOWN:
        assert R.crc32c(data, init) == want
        assert L.zvec_hip_crc32c(data, len(data), init) == want
    data = b"123456789\0"          # the reference hashes 10 bytes of a 9-char literal in its last iteration
    for i, want in enumerate(CRC_PREFIXES):
        assert R.crc32c(data[: i + 1]) == want
        assert L.zvec_hip_crc32c(data, i + 1, 0) == want


def test_roaring_writer_layout():
    # no run containers: cookie 12346, count, (key, card-1) pairs, offsets, payloads
    ids = [5, 7, 65536 + 9] + list(range(3 * 65536, 3 * 65536 + 5000))
    b = R.serialize32(ids)
    cookie, n = struct.unpack_from("<II", b, 0)
    assert cookie == 12346 and n == 3
    assert struct.unpack_from("<HHHHHH", b, 8) == (0, 1, 1, 0, 3, 4999)
    offs = struct.unpack_from("<III", b, 20)
    assert offs == (32, 36, 38) and len(b) == 38 + 8192          # two arrays, one bitset
    # run-optimised: cookie 12347 | (n-1) << 16, run flags, no offset header below 4 containers
    b = R.serialize32(list(range(100, 1100)) + [70000], run_optimize=True)
    assert struct.unpack_from("<I", b, 0)[0] == (12347 | (1 << 16)) and b[4] == 0b01
    assert struct.unpack_from("<HHHH", b, 5) == (0, 999, 1, 0)
    assert struct.unpack_from("<HHH", b, 13) == (1, 100, 999) and len(b) == 13 + 6 + 2
    # 64-bit map: bucket count, then (high, stream) pairs
    b = R.serialize64map([1, (7 << 32) + 3])
    assert struct.unpack_from("<Q", b, 0)[0] == 2 and struct.unpack_from("<I", b, 8)[0] == 0
    img = R.file_image(b, is_32bit=False, timestamp=42)
    assert len(img) == 64 + len(b) and struct.unpack_from("<QII", img, 0) == (R.FILE_MAGIC, 0, R.crc32c(b))


def test_doc_filter_mask_semantics():
    keys = np.array([0, 1, 2, 3, (1 << 32) + 1, 9], np.uint64)
    # 32-bit delete bitmap is probed with (uint32_t)id: key 2^32+1 aliases id 1 (concurrent_roaring_bitmap.h:196-203)
    assert R.doc_filter_mask(keys, deleted=[1]).tolist() == [False, True, False, False, True, False]
    assert R.doc_filter_mask(keys, deleted=[1], deleted_is32=False).tolist() == [False, True, False, False, False, False]
    assert R.doc_filter_mask(keys, invert=[0, 2, 9]).tolist() == [False, True, False, True, True, False]
    # forward bits beyond the array do not exclude (doc_filter.cc:104-107)
    assert R.doc_filter_mask(keys, forward=[True, False, True, True]).tolist() == [False, True, False, False, False, False]
    assert R.mask_to_words([True] + [False] * 63 + [True]).tolist() == [1, 1]
