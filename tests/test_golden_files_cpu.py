"""Golden index FILES dumped by the reference's own writers (tests/golden/ref_index_files.npz, made by
tests/golden/make_ref_index_files.py from FlatBuilder<32> / IVFDumper / MemoryDumper + IndexPacker compiled in place) —
CPU side: the container parser (host-only code of the C ABI) reads them, checksums included; and the test-side restated
writers (tests/ivf_format.py), which other tests feed the loaders with, reproduce the reference's segment payloads byte
for byte.  SURVEY §8(f) next-2: layout parity pinned."""
import os
import struct

import numpy as np
import pytest

from tests import ivf_format as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def golden():
    return np.load(os.path.join(ROOT, "tests", "golden", "ref_index_files.npz"))


def _case(z, name):
    meta = z[name + "_meta"]
    f16 = bool(meta[0])
    dt = np.float16 if f16 else np.float32
    d = {"image": z[name + "_image"].tobytes(), "base": z[name + "_base"].view(dt) if f16 else z[name + "_base"],
         "keys": z[name + "_keys"], "n": int(meta[1]), "dim": int(meta[2]), "column_major": bool(meta[3]), "dtype": dt}
    if name.startswith("ivf"):
        d.update(cent=z[name + "_cent"].view(dt) if f16 else z[name + "_cent"], offs=z[name + "_offs"],
                 centroid_column_major=bool(meta[4]), nlist=int(meta[5]))
    return d


def test_container_parser_reads_reference_dumped_files(golden):
    import zvec_amd
    from zvec_amd.index import container_segments, parse_index_meta
    for name in golden["cases"]:
        c = _case(golden, str(name))
        seg = container_segments(c["image"], checksum=True)          # header / footer / meta / content CRCs all verified
        want = {"IndexMeta", "IndexVersion"} | ({"flat.keys", "flat.features"} if name.startswith("flat") else
                                                {"ivf.centroid", "ivf.inverted_body", "ivf.inverted_header", "ivf.inverted_meta", "hc.keys"})
        assert want <= set(seg), (name, sorted(seg))
        for sid, (off, size) in seg.items():
            assert off % 32 == 0 and off + size <= len(c["image"])
        m = parse_index_meta(c["image"][seg["IndexMeta"][0]:sum(seg["IndexMeta"])])
        assert m["dimension"] == c["dim"] and m["metric"] == "InnerProduct"
        assert m["data_type"] == (1 if c["dtype"] == np.float16 else 2) and m["major_order"] == (2 if c["column_major"] else 1)
        # a flipped content byte is caught by the content checksum only
        bad = bytearray(c["image"])
        bad[64 + 3] ^= 0x10
        with pytest.raises(zvec_amd._lib.ZvecHipError):
            container_segments(bytes(bad), checksum=True)


def test_restated_writers_reproduce_reference_payloads(golden):
    """tests/ivf_format.py (FlatBuilder / IVFDumper / IndexMeta header restated) vs the reference's own output"""
    from zvec_amd.index import container_segments

    def seg_bytes(image, seg, sid):
        return image[seg[sid][0]:seg[sid][0] + seg[sid][1]]
    for name in golden["cases"]:
        name = str(name)
        c = _case(golden, name)
        seg = container_segments(c["image"])
        if name.startswith("flat"):
            assert seg_bytes(c["image"], seg, "flat.keys") == c["keys"].astype("<u8").tobytes()
            assert seg_bytes(c["image"], seg, "flat.features") == F.flat_features_blob(c["base"], c["column_major"])
        else:
            offs = c["offs"].astype(np.int64)
            lists = [(c["base"][offs[l]:offs[l + 1]], c["keys"][offs[l]:offs[l + 1]]) for l in range(c["nlist"])]
            mine = F.dump_ivf_segments(lists, c["dim"], c["dtype"], c["column_major"])
            # (the padding behind a list's last, partial block is whatever the dumper's reused block buffer held: zero it)
            ref_body = bytearray(seg_bytes(c["image"], seg, "ivf.inverted_body"))
            elem = c["dim"] * np.dtype(c["dtype"]).itemsize
            bsz = (32 * elem + 31) // 32 * 32
            pos = 0
            for l in range(c["nlist"]):
                cnt = int(offs[l + 1] - offs[l])
                pos += cnt // 32 * bsz
                if cnt % 32:
                    used, padded = (cnt % 32) * elem, ((cnt % 32) * elem + 31) // 32 * 32
                    ref_body[pos + used:pos + padded] = b"\0" * (padded - used)
                    pos += padded
            assert bytes(ref_body) == mine["ivf.inverted_body"], name
            assert seg_bytes(c["image"], seg, "ivf.inverted_meta") == mine["ivf.inverted_meta"], name
            assert seg_bytes(c["image"], seg, "hc.keys") == mine["hc.keys"], name
            ref_hdr = seg_bytes(c["image"], seg, "ivf.inverted_header")
            # InvertedIndexHeader: everything but header_size / index_meta_size (the reference's embedded IndexMeta carries
            # its JSON attachment, the test writer's does not)
            assert ref_hdr[4:32] == mine["ivf.inverted_header"][4:32]
            assert struct.unpack_from("<I", ref_hdr, 0)[0] == len(ref_hdr) and struct.unpack_from("<I", ref_hdr, 32)[0] == len(ref_hdr) - 64
            assert struct.unpack_from("<9I", ref_hdr, 64)[1:6] == struct.unpack_from("<9I", mine["ivf.inverted_header"], 64)[1:6]
            # the nested centroid index is itself a dumped flat index file
            nested = seg_bytes(c["image"], seg, "ivf.centroid")
            cseg = container_segments(nested, checksum=True)
            assert seg_bytes(nested, cseg, "flat.features") == F.flat_features_blob(c["cent"], c["centroid_column_major"])
            assert np.array_equal(np.frombuffer(seg_bytes(nested, cseg, "flat.keys"), np.uint64), np.arange(c["nlist"], dtype=np.uint64))
        # the restated packer: same framing as the reference's for the same segments (ids, sizes, padding, order)
        order = sorted(seg, key=lambda s: seg[s][0])
        payloads = [(sid, seg_bytes(c["image"], seg, sid)) for sid in order if sid != "IndexVersion"]
        version = seg_bytes(c["image"], seg, "IndexVersion")
        mine = F.pack_container(payloads, version=version)
        mseg = container_segments(mine, checksum=True)
        assert mseg == seg and len(mine) == len(c["image"])
        # the whole content area is identical (the meta block differs only in the per-segment data_crc, which the
        # reference's builders leave 0 for these segments, and the header / footer in time stamps, magic and their CRCs)
        meta_size = struct.unpack_from("<I", c["image"], len(c["image"]) - 128 + 16)[0]
        content_end = len(c["image"]) - 128 - meta_size
        assert mine[64:content_end] == c["image"][64:content_end]
        assert struct.unpack_from("<I", mine, len(mine) - 128 + 8)[0] == struct.unpack_from("<I", c["image"], len(mine) - 128 + 8)[0]   # content_crc
