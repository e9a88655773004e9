"""CPU: .flo codec, KITTI uint16 arithmetic and the evaluation metrics (reference frame_utils.py / evaluate_mf.py)."""
import os

import numpy as np
import pytest

from streamflow_amd import flow_io


def test_flo_round_trip_and_header_bytes(tmp_path):
    rng = np.random.default_rng(0)
    flow = rng.standard_normal((5, 7, 2)).astype(np.float32) * 10
    p = tmp_path / "a.flo"
    flow_io.write_flo(str(p), flow)
    raw = p.read_bytes()
    assert raw[:4] == b"PIEH"                                  # 202021.25f, the Middlebury tag
    assert np.frombuffer(raw[4:12], np.int32).tolist() == [7, 5]   # width first, then height
    assert len(raw) == 12 + 5 * 7 * 2 * 4
    assert np.frombuffer(raw[12:20], np.float32).tolist() == flow[0, 0].tolist()   # (u, v) interleaved
    assert np.array_equal(flow_io.read_flo(str(p)), flow)
    p.write_bytes(b"XXXX" + raw[4:])
    with pytest.raises(IOError):
        flow_io.read_flo(str(p))


def test_kitti_codec_arithmetic():
    flow = np.array([[[1.5, -2.25], [0.0, 100.0]]], np.float32)          # multiples of 1/64 are exact
    enc = flow_io.kitti_encode(flow)
    assert enc.dtype == np.uint16 and enc[0, 0].tolist() == [32768 + 96, 32768 - 144, 1]
    dec, valid = flow_io.kitti_decode(enc)
    assert np.array_equal(dec, flow) and valid.tolist() == [[1.0, 1.0]]


def test_metrics_formulas():
    gt = np.zeros((2, 2, 3))
    gt[0] = 10.0                                            # |gt| = 10
    flow = gt.copy()
    flow[0, 0, 0] += 3.0                                    # epe 3   -> not > 3: inlier
    flow[0, 0, 1] += 4.0                                    # epe 4, 40 % -> outlier
    flow[1, 1, 2] += 0.5                                    # epe 0.5
    e = flow_io.epe_map(flow, gt)
    assert np.allclose(e, [[3, 4, 0], [0, 0, 0.5]])
    m = flow_io.sintel_metrics(e)
    assert m["epe"] == pytest.approx(7.5 / 6) and m["1px"] == pytest.approx(4 / 6) and m["5px"] == 1.0
    valid = np.ones((2, 3))
    valid[0, 0] = 0
    k = flow_io.kitti_f1(flow, gt, valid)
    assert k["f1"] == pytest.approx(100 * 1 / 5) and k["epe"] == pytest.approx(4.5 / 5)
    # HWC layout gives the same numbers
    assert np.allclose(flow_io.epe_map(flow.transpose(1, 2, 0), gt.transpose(1, 2, 0)), e)


def test_kitti_png_round_trip_and_filters(tmp_path):
    """16-bit KITTI flow PNG (frame_utils.py:117-122,137-141): write -> read is exact on the 1/64 px grid, the file is a
    16-bit RGB PNG with u in the RED channel (cv2's BGR flip in the reference), and the reader decodes all five PNG row
    filters (a hand-filtered file: devkit PNGs are written by libpng with adaptive filtering)."""
    import struct
    import zlib
    from streamflow_amd import flow_io
    rng = np.random.default_rng(5)
    flow = np.round(rng.normal(0, 30, (13, 17, 2)) * 64) / 64
    path = str(tmp_path / "f.png")
    flow_io.write_flow_kitti(path, flow)
    raw = open(path, "rb").read()
    assert raw[:8] == b"\x89PNG\r\n\x1a\n" and struct.unpack(">IIBB", raw[16:26]) == (17, 13, 16, 2)
    got, valid = flow_io.read_flow_kitti(path)
    assert got.dtype == np.float32 and np.array_equal(got, flow.astype(np.float32)) and (valid == 1).all()
    img = flow_io.read_png(path)
    assert np.array_equal(img[:, :, 0], (64 * flow[:, :, 0] + 2 ** 15).astype(np.uint16))       # R = u
    # re-encode the same pixels with filter types 0..4 cycling over the rows
    rows = img.astype(">u2").reshape(13, -1).view(np.uint8).reshape(13, -1).astype(np.int32)
    bpp, enc, prev = 6, [], np.zeros(rows.shape[1], np.int32)
    for y, line in enumerate(rows):
        ft = y % 5
        a = np.concatenate([np.zeros(bpp, np.int32), line[:-bpp]])
        c = np.concatenate([np.zeros(bpp, np.int32), prev[:-bpp]])
        p = a + prev - c
        pa, pb, pc = np.abs(p - a), np.abs(p - prev), np.abs(p - c)
        paeth = np.where((pa <= pb) & (pa <= pc), a, np.where(pb <= pc, prev, c))
        pred = [0, a, prev, (a + prev) >> 1, paeth][ft]
        enc.append(bytes([ft]) + ((line - pred) & 255).astype(np.uint8).tobytes())
        prev = line

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)
    p2 = str(tmp_path / "g.png")
    open(p2, "wb").write(raw[:8] + chunk(b"IHDR", struct.pack(">IIBBBBB", 17, 13, 16, 2, 0, 0, 0)) +
                         chunk(b"IDAT", zlib.compress(b"".join(enc))[:40]) + chunk(b"IDAT", zlib.compress(b"".join(enc))[40:]) +
                         chunk(b"IEND", b""))
    assert np.array_equal(flow_io.read_png(p2), img)
    got2, _ = flow_io.read_flow_kitti(p2)
    assert np.array_equal(got2, got)
    with pytest.raises(IOError):
        flow_io.read_png(str(tmp_path / "f.png") + "x") if False else flow_io.read_flow_kitti(_gray_png(tmp_path, flow_io))


def _gray_png(tmp_path, flow_io):
    p = str(tmp_path / "gray.png")
    flow_io.write_png(p, np.zeros((4, 4), np.uint8))
    return p


def test_pfm_round_trip(tmp_path):
    from streamflow_amd import flow_io
    rng = np.random.default_rng(6)
    for shape in ((5, 7), (5, 7, 3)):
        a = rng.normal(size=shape).astype(np.float32)
        p = str(tmp_path / "a.pfm")
        flow_io.write_pfm(p, a)
        assert np.array_equal(flow_io.read_pfm(p), a)
    # big-endian file with a positive scale, rows bottom-up (frame_utils.py:70-81)
    b = rng.normal(size=(3, 4)).astype(np.float32)
    p = str(tmp_path / "b.pfm")
    with open(p, "wb") as f:
        f.write(b"Pf\n4 3\n1.0\n")
        np.flipud(b).astype(">f4").tofile(f)
    assert np.array_equal(flow_io.read_pfm(p), b)


def test_flo5_round_trip_and_structures(tmp_path):
    """Spring's .flo5 (frame_utils.py:31-47,130-135): an HDF5 file with one gzip dataset 'flow'.  Round trips at ragged
    sizes and chunkings, and the on-disk structures are checked field by field against the HDF5 format specification
    (superblock v0, symbol-table group, v1 object header, chunked layout v3, deflate filter)."""
    import struct
    import zlib
    from streamflow_amd import flow_io
    from streamflow_amd.flo5 import read_hdf5_dataset
    rng = np.random.default_rng(5)
    for (h, w, rpc) in [(436, 1024, 0), (1, 1, 0), (37, 53, 5), (1080, 64, 17), (70, 9, 70)]:
        flow = (rng.standard_normal((h, w, 2)) * 20).astype(np.float32)
        p = str(tmp_path / f"f_{h}_{w}.flo5")
        flow_io.write_flo5(p, flow, rows_per_chunk=rpc)
        back = flow_io.read_flo5(p)
        assert back.dtype == np.float32 and back.shape == (h, w, 2) and np.array_equal(back, flow)
    raw = open(p, "rb").read()
    assert raw[:8] == b"\x89HDF\r\n\x1a\n" and raw[8] == 0 and raw[13] == 8 and raw[14] == 8
    leaf_k, int_k = struct.unpack_from("<HH", raw, 16)
    base, fs, eof, drv = struct.unpack_from("<4Q", raw, 24)
    assert (leaf_k, int_k, base, eof) == (4, 16, 0, len(raw)) and fs == drv == 2 ** 64 - 1
    name_off, root_hdr, cache, _, btree, heap = struct.unpack_from("<QQIIQQ", raw, 56)
    assert cache == 1 and raw[btree:btree + 4] == b"TREE" and raw[heap:heap + 4] == b"HEAP"
    ver, nmsg, refs, size = struct.unpack_from("<BxHII", raw, root_hdr)
    assert (ver, nmsg, refs) == (1, 1, 1) and struct.unpack_from("<HH", raw, root_hdr + 16) == (0x11, 16)
    snod = struct.unpack_from("<Q", raw, btree + 32)[0]
    assert raw[snod:snod + 4] == b"SNOD" and struct.unpack_from("<H", raw, snod + 6)[0] == 1
    noff, dset = struct.unpack_from("<QQ", raw, snod + 8)
    heap_data = struct.unpack_from("<Q", raw, heap + 24)[0]
    assert raw[heap_data + noff:heap_data + noff + 5] == b"flow\0"
    # dataset header: dataspace, datatype (IEEE little-endian float32), fill value, filter pipeline (deflate 5), layout
    ver, nmsg, _, size = struct.unpack_from("<BxHII", raw, dset)
    pos, seen = dset + 16, {}
    for _ in range(nmsg):
        t, n = struct.unpack_from("<HH", raw, pos)
        seen[t] = raw[pos + 8:pos + 8 + n]
        pos += 8 + n
    assert pos == dset + 16 + size and set(seen) == {0x1, 0x3, 0x5, 0xB, 0x8}
    assert seen[0x1][:2] == b"\x01\x03" and struct.unpack_from("<3Q", seen[0x1], 8) == (70, 9, 2)
    assert seen[0x3][:8] == bytes([0x11, 0x20, 0x1F, 0x00, 4, 0, 0, 0])
    assert struct.unpack_from("<HHBBBBI", seen[0x3], 8) == (0, 32, 23, 8, 0, 23, 127)
    assert struct.unpack_from("<HHHH", seen[0xB], 8) == (1, 8, 1, 1) and struct.unpack_from("<I", seen[0xB], 24)[0] == 5
    lver, lclass, nd = seen[0x8][:3]
    cb = struct.unpack_from("<Q", seen[0x8], 3)[0]
    assert (lver, lclass, nd) == (3, 2, 4) and struct.unpack_from("<4I", seen[0x8], 11) == (70, 9, 2, 4)
    sig, ntype, level, used = struct.unpack_from("<4sBBH", raw, cb)
    assert (sig, ntype, level, used) == (b"TREE", 1, 0, 1)
    nbytes, mask, o0, o1, o2, o3, addr = struct.unpack_from("<II4QQ", raw, cb + 24)
    assert (mask, o0, o1, o2, o3) == (0, 0, 0, 0, 0) and addr + nbytes == len(raw)
    assert np.array_equal(np.frombuffer(zlib.decompress(raw[addr:addr + nbytes]), "<f4").reshape(70, 9, 2), flow)
    with pytest.raises(IOError, match="does not have a 'nope' key"):
        read_hdf5_dataset(p, "nope")
    bad = str(tmp_path / "bad.flo5")
    open(bad, "wb").write(b"not hdf5" * 100)
    with pytest.raises(IOError, match="not an HDF5 file"):
        flow_io.read_flo5(bad)
    with pytest.raises(ValueError):
        flow_io.write_flo5(bad, np.zeros((4, 4, 3), np.float32))


def test_flo5_reader_handles_other_layouts(tmp_path):
    """The reader on structures h5py may also emit: contiguous layout, a shuffle + deflate pipeline, a two-level chunk
    B-tree, big-endian doubles -- files assembled here from the specification's field tables."""
    import struct
    import zlib
    from streamflow_amd import flo5
    U = flo5.UNDEF

    def build(dset_msgs, extra):
        # superblock | root header | group B-tree | heap | heap data | SNOD | dataset header | extra blobs
        off_root, off_bt = 96, 96 + 40
        off_heap = off_bt + 544
        off_hd = off_heap + 32
        off_snod = off_hd + 32
        off_dset = off_snod + 328
        hdr = flo5._object_header(dset_msgs(off_dset))
        blob = extra(off_dset + len(hdr))
        eof = off_dset + len(hdr) + len(blob)
        sup = (flo5.SIGNATURE + bytes([0, 0, 0, 0, 0, 8, 8, 0]) + struct.pack("<HHI", 4, 16, 0) +
               struct.pack("<4Q", 0, U, eof, U) + struct.pack("<QQII", 0, off_root, 1, 0) + struct.pack("<QQ", off_bt, off_heap))
        root = flo5._object_header([flo5._message(0x11, struct.pack("<QQ", off_bt, off_heap))])
        gb = (struct.pack("<4sBBHQQ", b"TREE", 0, 0, 1, U, U) + struct.pack("<QQQ", 0, off_snod, 8)).ljust(544, b"\0")
        heap = struct.pack("<4sB3xQQQ", b"HEAP", 0, 32, 16, off_hd)
        hd = b"\0" * 8 + b"flow\0\0\0\0" + struct.pack("<QQ", 1, 16)
        snod = (struct.pack("<4sBxH", b"SNOD", 1, 1) + struct.pack("<QQII16x", 8, off_dset, 0, 0)).ljust(328, b"\0")
        return sup + root + gb + heap + hd + snod + hdr + blob

    data = (np.arange(6 * 5 * 2, dtype=np.float64).reshape(6, 5, 2) - 17.5)
    space = struct.pack("<BBB5x", 1, 3, 0) + struct.pack("<3Q", 6, 5, 2)
    be_f64 = struct.pack("<B3BI", 0x11, 0x21, 0x3F, 0, 8) + struct.pack("<HHBBBBI", 0, 64, 52, 11, 0, 52, 1023)
    # (1) contiguous, big-endian float64
    raw = build(lambda d: [flo5._message(1, space), flo5._message(3, be_f64),
                           flo5._message(8, struct.pack("<BB", 3, 1) + struct.pack("<QQ", d + 16 + 40 + 32 + 32, 480))],
                lambda a: data.astype(">f8").tobytes())
    p = str(tmp_path / "contig.h5")
    open(p, "wb").write(raw)
    got = flo5.read_flo5(p)
    assert got.dtype == np.dtype(">f8") and np.array_equal(got, data)
    # (2) chunks of [4, 5, 2] float32 behind a two-level B-tree, pipeline shuffle -> deflate, second chunk with the
    #     deflate stage skipped (filter mask bit 1)
    d32 = data.astype("<f4")
    space32 = space
    f32 = flo5._float32_datatype()
    pipeline = (struct.pack("<BB6x", 1, 2) + struct.pack("<HHHH", 2, 0, 1, 1) + struct.pack("<I4x", 4) +
                struct.pack("<HHHH", 1, 0, 1, 1) + struct.pack("<I4x", 6))

    def shuffled(a):
        b = np.zeros((4, 5, 2), "<f4")
        b[: a.shape[0]] = a
        return np.frombuffer(b.tobytes(), np.uint8).reshape(-1, 4).T.tobytes()

    c0, c1 = zlib.compress(shuffled(d32[:4]), 6), shuffled(d32[4:])
    ksz = 8 + 8 * 4

    def tree(a):                                            # a = address right after the dataset header
        root_sz = 24 + 64 * 8 + 65 * ksz
        leaf = a + root_sz
        chunks = leaf + root_sz
        lf = struct.pack("<4sBBHQQ", b"TREE", 1, 0, 2, U, U)
        lf += struct.pack("<II4Q", len(c0), 0, 0, 0, 0, 0) + struct.pack("<Q", chunks)
        lf += struct.pack("<II4Q", len(c1), 2, 4, 0, 0, 0) + struct.pack("<Q", chunks + len(c0))
        lf += struct.pack("<II4Q", 0, 0, 8, 0, 0, 0)
        rt = struct.pack("<4sBBHQQ", b"TREE", 1, 1, 1, U, U)
        rt += struct.pack("<II4Q", len(c0), 0, 0, 0, 0, 0) + struct.pack("<Q", leaf) + struct.pack("<II4Q", 0, 0, 8, 0, 0, 0)
        return rt.ljust(root_sz, b"\0") + lf.ljust(root_sz, b"\0") + c0 + c1

    def msgs(d):
        m = [flo5._message(1, space32), flo5._message(3, f32), flo5._message(0xB, pipeline)]
        size = 16 + sum(len(x) for x in m) + 8 + 32
        return m + [flo5._message(8, struct.pack("<BBB", 3, 2, 4) + struct.pack("<Q", d + size) + struct.pack("<4I", 4, 5, 2, 4))]

    p2 = str(tmp_path / "chunked.h5")
    open(p2, "wb").write(build(msgs, tree))
    got = flo5.read_flo5(p2)
    assert got.dtype == np.float32 and np.array_equal(got, d32)


def test_flow_formats_pinned_against_reference_frame_utils(golden, tmp_path):
    """f3 pinned: .flo, PFM and the KITTI uint16 arithmetic against outputs of the reference's own core/utils/frame_utils.py
    (readFlow / writeFlow / readPFM / readFlowKITTI / writeFlowKITTI, executed over cv2 / h5py stubs by
    tests/golden/make_golden.py).  The PNG container is cv2's in the reference and our own zlib codec here: what is
    pinned for KITTI is the array handed to / received from the codec."""
    import numpy as np
    from streamflow_amd import flow_io
    from tests import cases
    g = golden("frame_utils")
    flow, kitti, pfm3, pfm1 = cases.flow_io_inputs()
    # .flo: byte-identical file, identical read-back
    p = tmp_path / "a.flo"
    flow_io.write_flo(str(p), flow)
    assert np.array_equal(np.fromfile(p, np.uint8), g["flo_bytes"])
    g["flo_bytes"].tofile(tmp_path / "ref.flo")
    assert np.array_equal(flow_io.read_flo(str(tmp_path / "ref.flo")), g["flo_back"])
    # KITTI encode: the reference passes uv[..., ::-1] (B, G, R = valid, v, u) to cv2.imwrite; ours holds (u, v, valid)
    assert np.array_equal(flow_io.kitti_encode(flow)[..., ::-1], g["kitti_written_bgr"])
    # ... through our PNG codec and back: the same codes
    flow_io.write_flow_kitti(str(tmp_path / "k.png"), flow)
    assert np.array_equal(flow_io.read_png(str(tmp_path / "k.png"))[..., ::-1], g["kitti_written_bgr"])
    # KITTI decode: cv2.imread returns B, G, R; the file holds R, G, B = (u, v, valid)
    flow_io.write_png(str(tmp_path / "in.png"), np.ascontiguousarray(kitti[..., ::-1]))
    f, v = flow_io.read_flow_kitti(str(tmp_path / "in.png"))
    assert f.dtype == np.float32 and np.array_equal(f, g["kitti_flow"]) and np.array_equal(v, g["kitti_valid"])
    # PFM: the reference's reader on little-endian colour and big-endian grey files
    for tag in ("pfm3_le", "pfm1_be"):
        g[tag + "_file"].tofile(tmp_path / (tag + ".pfm"))
        assert np.array_equal(flow_io.read_pfm(str(tmp_path / (tag + ".pfm"))), g[tag])
    flow_io.write_pfm(str(tmp_path / "w.pfm"), pfm3)
    assert np.array_equal(np.fromfile(tmp_path / "w.pfm", np.uint8), g["pfm3_le_file"])


def test_png_reader_all_filter_types(tmp_path):
    """libpng writes adaptive filters (Sub / Up / Average / Paeth): a 16-bit RGB image filtered by hand, every type in turn,
    must decode exactly; a corrupted chunk CRC is rejected."""
    import struct, zlib
    import numpy as np
    from streamflow_amd import flow_io
    rng = np.random.default_rng(0)
    h, w, bpp = 23, 31, 6
    img = rng.integers(0, 65536, size=(h, w, 3)).astype(np.uint16)
    rows = img.astype(">u2").reshape(h, -1).view(np.uint8).reshape(h, -1).astype(np.int32)
    raw, prev = bytearray(), np.zeros(w * bpp, np.int32)
    for y in range(h):
        ft, cur, line = y % 5, rows[y], np.zeros(w * bpp, np.int32)
        for x in range(w * bpp):
            a, b, c = (cur[x - bpp] if x >= bpp else 0), prev[x], (prev[x - bpp] if x >= bpp else 0)
            p = a + b - c
            pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
            pred = [0, a, b, (a + b) >> 1, a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)][ft]
            line[x] = (cur[x] - pred) & 255
        raw.append(ft)
        raw += bytes(line.astype(np.uint8))
        prev = cur
    chunk = lambda tag, data: struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)
    png = (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 16, 2, 0, 0, 0)) +
           chunk(b"IDAT", zlib.compress(bytes(raw))) + chunk(b"IEND", b""))
    (tmp_path / "f.png").write_bytes(png)
    assert np.array_equal(flow_io.read_png(str(tmp_path / "f.png")), img)
    bad = bytearray(png)
    bad[60] ^= 0x40
    (tmp_path / "bad.png").write_bytes(bytes(bad))
    with pytest.raises(IOError):
        flow_io.read_png(str(tmp_path / "bad.png"))


GOLD5 = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "flo5")
H5PY_PYTHON = "/opt/conda/bin/python3.9"            # the build container's interpreter that has h5py 3.3.0 / HDF5 1.10.6


def test_flo5_reader_against_files_written_by_h5py():
    """Parity pin of the .flo5 reader: the fixtures under tests/golden/flo5 were written by the real h5py with the reference's
    own call (frame_utils.py:46-47, gzip level 5, automatic chunking; plus contiguous float64 and shuffled custom chunks) by
    tests/golden/make_flo5_golden.py -- bytes this package did not produce.  Bit-exact, NaNs (invalid pixels) included."""
    from streamflow_amd import flow_io
    exp = np.load(os.path.join(GOLD5, "expected.npz"))
    assert len(exp.files) >= 6
    for name in exp.files:
        got = flow_io.read_flo5(os.path.join(GOLD5, name + ".flo5"))
        assert got.dtype == exp[name].dtype and got.shape == exp[name].shape, name
        assert np.array_equal(got.view(np.uint8), exp[name].view(np.uint8)), name          # bit patterns, NaN payloads too


@pytest.mark.skipif(not os.path.exists(H5PY_PYTHON), reason="needs the build container's h5py interpreter")
def test_flo5_writer_output_opens_in_h5py(tmp_path):
    """The other direction: files from write_flo5 are read back by the real libhdf5 (same values, gzip level 5, chunked)."""
    import subprocess
    from streamflow_amd import flow_io
    rng = np.random.default_rng(5)
    paths, sums = [], []
    for i, (h, w) in enumerate(((1, 1), (37, 53), (136, 240))):
        flow = (rng.standard_normal((h, w, 2)) * 9).astype(np.float32)
        if h > 1:
            flow[rng.random((h, w)) < 0.05] = np.nan
        p = str(tmp_path / f"w{i}.flo5")
        flow_io.write_flo5(p, flow)
        paths.append(p)
        sums.append(float(np.nansum(flow.astype(np.float64))))
    code = ("import sys, h5py, numpy as np\n"
            "for p in sys.argv[1:]:\n"
            "    with h5py.File(p, 'r') as f:\n"
            "        d = f['flow']; a = d[()]\n"
            "        print(a.shape[0], a.shape[1], a.shape[2], a.dtype, d.compression, d.compression_opts, repr(float(np.nansum(a.astype(np.float64)))))\n")
    out = subprocess.run([H5PY_PYTHON, "-c", code] + paths, capture_output=True, text=True, timeout=120,
                         env={"PATH": "/usr/bin:/bin"})
    assert out.returncode == 0, out.stderr[-500:]
    lines = out.stdout.strip().splitlines()
    assert len(lines) == 3
    for line, (h, w), s in zip(lines, ((1, 1), (37, 53), (136, 240)), sums):
        f = line.split()
        assert (int(f[0]), int(f[1]), int(f[2])) == (h, w, 2) and f[3] == "float32" and f[4] == "gzip" and f[5] == "5"
        assert float(f[6]) == s
