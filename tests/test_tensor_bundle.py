"""TensorBundle (TF checkpoint V2) writer/reader.

TensorFlow is not installable here, so parity is pinned two ways: (1) an independent pure-Python reader of
the on-disk format (LevelDB table + BundleHeader/BundleEntry protos + masked crc32c) written straight from
the format description, and (2) hand-derived golden bytes for a tiny checkpoint.  Cross-reading with a real
TensorFlow build is "parity unpinned".
"""
import os
import struct

import pytest
import torch

from distributed_tensorflow_amd import saver as sv

# ----------------------------------------------------------------------------- independent pure-python reader
_CRC_TABLE = []
for _i in range(256):
    _c = _i
    for _ in range(8):
        _c = (_c >> 1) ^ 0x82F63B78 if _c & 1 else _c >> 1
    _CRC_TABLE.append(_c)


def crc32c(data: bytes) -> int:
    c = 0xFFFFFFFF
    for b in data:
        c = _CRC_TABLE[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def mask(crc: int) -> int:
    return ((((crc >> 15) | (crc << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


def varint(b, i):
    r = s = 0
    while True:
        c = b[i]
        i += 1
        r |= (c & 0x7F) << s
        if not c & 0x80:
            return r, i
        s += 7


def read_block(f: bytes, off: int, size: int) -> bytes:
    body, typ = f[off:off + size], f[off + size]
    (stored,) = struct.unpack_from("<I", f, off + size + 1)
    assert typ == 0
    assert stored == mask(crc32c(body + bytes([typ])))
    return body


def block_entries(block: bytes):
    (nrestarts,) = struct.unpack_from("<I", block, len(block) - 4)
    end = len(block) - 4 - 4 * nrestarts
    restarts = struct.unpack_from("<%dI" % nrestarts, block, end)
    assert restarts[0] == 0
    i, key, out = 0, b"", []
    while i < end:
        shared, i = varint(block, i)
        non_shared, i = varint(block, i)
        vlen, i = varint(block, i)
        key = key[:shared] + block[i:i + non_shared]
        i += non_shared
        out.append((key, block[i:i + vlen]))
        i += vlen
    return out, restarts


def parse_proto(b: bytes):
    i, out = 0, []
    while i < len(b):
        k, i = varint(b, i)
        f, w = k >> 3, k & 7
        if w == 0:
            v, i = varint(b, i)
        elif w == 2:
            n, i = varint(b, i)
            v = b[i:i + n]
            i += n
        elif w == 5:
            (v,) = struct.unpack_from("<I", b, i)
            i += 4
        else:
            raise AssertionError("unexpected wire type %d" % w)
        out.append((f, w, v))
    return out


def py_read_index(path: str):
    f = open(path, "rb").read()
    footer = f[-48:]
    assert struct.unpack("<Q", footer[40:])[0] == 0xDB4775248B80FB57
    mo, i = varint(footer, 0)
    ms, i = varint(footer, i)
    io, i = varint(footer, i)
    isz, i = varint(footer, i)
    assert block_entries(read_block(f, mo, ms))[0] == []       # empty metaindex
    index_entries, _ = block_entries(read_block(f, io, isz))
    entries = []
    for sep, handle in index_entries:
        o, j = varint(handle, 0)
        s, j = varint(handle, j)
        blk_entries, _ = block_entries(read_block(f, o, s))
        assert blk_entries[-1][0] <= sep                      # separator >= last key of its block
        entries.extend(blk_entries)
    keys = [k for k, _ in entries]
    assert keys == sorted(keys) and keys[0] == b""
    header = dict((fld, v) for fld, _, v in parse_proto(entries[0][1]))
    out = {}
    for k, v in entries[1:]:
        e = {"dtype": 0, "shape": [], "shard_id": 0, "offset": 0, "size": 0, "crc32c": 0}
        for fld, w, val in parse_proto(v):
            if fld == 1: e["dtype"] = val
            elif fld == 2:
                for f2, _, dim in parse_proto(val):
                    assert f2 == 2
                    e["shape"].append(dict((a, c) for a, _, c in parse_proto(dim)).get(1, 0))
            elif fld == 3: e["shard_id"] = val
            elif fld == 4: e["offset"] = val
            elif fld == 5: e["size"] = val
            elif fld == 6:
                assert w == 5
                e["crc32c"] = val
        out[k.decode()] = e
    return header, out


# ----------------------------------------------------------------------------- tests
def test_crc32c_check_value():
    from distributed_tensorflow_amd.summary import _masked_crc
    assert crc32c(b"123456789") == 0xE3069283
    assert _masked_crc(b"123456789") == mask(0xE3069283)


def test_golden_bytes_single_tensor(tmp_path):
    """One int64 scalar named 'global_step' = 7: every byte of both files is derivable by hand."""
    prefix = str(tmp_path / "model.ckpt-7")
    w = sv.BundleWriter(prefix, 1)
    w.add("global_step", torch.tensor(7, dtype=torch.int64))
    w.finish()
    data = open(prefix + ".data-00000-of-00001", "rb").read()
    assert data == struct.pack("<q", 7)
    header_val = bytes([0x08, 0x01, 0x1A, 0x02, 0x08, 0x01])                       # num_shards=1, version{producer=1}
    entry_val = bytes([0x08, 0x09, 0x12, 0x00, 0x28, 0x08, 0x35]) + struct.pack("<I", mask(crc32c(data)))
    block = (bytes([0, 0, len(header_val)]) + header_val +
             bytes([0, 11, len(entry_val)]) + b"global_step" + entry_val +
             struct.pack("<II", 0, 1))
    trailer = lambda b: bytes([0]) + struct.pack("<I", mask(crc32c(b + bytes([0]))))  # noqa: E731
    meta = struct.pack("<II", 0, 1)
    handle = lambda off, size: bytes([off]) + bytes([size])                           # noqa: E731  (both < 128)
    index = bytes([0, 1, 2]) + b"h" + handle(0, len(block)) + struct.pack("<II", 0, 1)  # short successor of "global_step"
    meta_off = len(block) + 5
    index_off = meta_off + len(meta) + 5
    footer = (handle(meta_off, len(meta)) + handle(index_off, len(index))).ljust(40, b"\0") + \
        struct.pack("<Q", 0xDB4775248B80FB57)
    expected = block + trailer(block) + meta + trailer(meta) + index + trailer(index) + footer
    assert open(prefix + ".index", "rb").read() == expected


def test_roundtrip_and_independent_reader(tmp_path):
    torch.manual_seed(0)
    tensors = {"dense_1/kernel": torch.randn(64, 128), "dense_1/bias": torch.zeros(128),
               "dense_1/kernel/Adam": torch.randn(64, 128), "global_step": torch.tensor(30000, dtype=torch.int64),
               "beta1_power": torch.tensor(0.5), "flags/bool": torch.tensor([True, False]),
               "half": torch.randn(3, 5).half(), "i32": torch.arange(6, dtype=torch.int32).view(2, 3)}
    for i in range(40):   # > 16 keys: exercises restart points and prefix compression
        tensors["layer_%02d/w" % i] = torch.full((i + 1,), float(i))
    prefix = str(tmp_path / "model.ckpt-30000")
    w = sv.BundleWriter(prefix, 1)
    for k, v in tensors.items():
        w.add(k, v)
    w.finish()

    with sv.BundleReader(prefix) as r:
        assert set(r.entries) == set(tensors)
        for k, v in tensors.items():
            got = r.get_tensor(k)
            assert got.dtype == v.dtype and tuple(got.shape) == tuple(v.shape)
            assert torch.equal(got, v)

    header, entries = py_read_index(prefix + ".index")
    assert header[1] == 1 and dict((f, v) for f, _, v in parse_proto(header[3]))[1] == 1
    raw = open(prefix + ".data-00000-of-00001", "rb").read()
    assert sum(e["size"] for e in entries.values()) == len(raw)          # back to back, no padding
    k = entries["dense_1/kernel"]
    assert k["dtype"] == 1 and k["shape"] == [64, 128] and k["size"] == 64 * 128 * 4
    blob = raw[k["offset"]:k["offset"] + k["size"]]
    assert k["crc32c"] == mask(crc32c(blob))
    assert torch.equal(torch.frombuffer(bytearray(blob), dtype=torch.float32).view(64, 128), tensors["dense_1/kernel"])
    assert entries["global_step"]["dtype"] == 9 and entries["global_step"]["shape"] == []
    assert sv.list_variables(prefix)[0][0] == "beta1_power"


def test_sharded_save_and_merge(tmp_path):
    prefix = str(tmp_path / "model.ckpt-5")
    a, b = torch.randn(7, 3), torch.randn(11)
    w0 = sv.BundleWriter(prefix, 2, only_shard=0)
    w0.add("a", a, 0)
    w0.finish()
    w1 = sv.BundleWriter(prefix, 2, only_shard=1)
    w1.add("b", b, 1)
    w1.finish()
    assert not os.path.exists(prefix + ".index")
    sv.merge_shards(prefix, 2)
    assert sorted(os.listdir(str(tmp_path))) == ["model.ckpt-5.data-00000-of-00002", "model.ckpt-5.data-00001-of-00002",
                                                 "model.ckpt-5.index"]
    header, entries = py_read_index(prefix + ".index")
    assert header[1] == 2 and entries["a"]["shard_id"] == 0 and entries["b"]["shard_id"] == 1
    with sv.BundleReader(prefix) as r:
        assert r.num_shards == 2 and torch.equal(r.get_tensor("a"), a) and torch.equal(r.get_tensor("b"), b)


def test_corruption_is_detected(tmp_path):
    prefix = str(tmp_path / "c")
    w = sv.BundleWriter(prefix, 1)
    w.add("x", torch.arange(100, dtype=torch.float32))
    w.finish()
    p = prefix + ".data-00000-of-00001"
    raw = bytearray(open(p, "rb").read())
    raw[17] ^= 0x40
    open(p, "wb").write(bytes(raw))
    with sv.BundleReader(prefix) as r:
        with pytest.raises(IOError):
            r.get_tensor("x")
        assert r.get_tensor("x", verify_crc=False).shape == (100,)
    idx = bytearray(open(prefix + ".index", "rb").read())
    idx[3] ^= 0x01
    open(prefix + ".index", "wb").write(bytes(idx))
    with pytest.raises(IOError):
        sv.BundleReader(prefix)


def test_saver_state_file_and_max_to_keep(tmp_path):
    from distributed_tensorflow_amd.models import xor_mlp
    m = xor_mlp()
    m.compile(loss="mean_squared_error", optimizer="adam", device="cpu", seed=0)
    s = sv.Saver(m.store, m.optimizer, max_to_keep=2)
    d = str(tmp_path)
    for step in (10, 20, 30):
        m.store.global_step.fill_(step)
        s.save(os.path.join(d, "model.ckpt"), global_step=step)
    text = open(os.path.join(d, "checkpoint")).read().splitlines()
    assert text == ['model_checkpoint_path: "model.ckpt-30"', 'all_model_checkpoint_paths: "model.ckpt-20"',
                    'all_model_checkpoint_paths: "model.ckpt-30"']
    assert not os.path.exists(os.path.join(d, "model.ckpt-10.index"))
    assert sv.latest_checkpoint(d) == os.path.join(d, "model.ckpt-30")
    names = dict(sv.list_variables(sv.latest_checkpoint(d)))
    expect = {"dense_1/kernel": (64, 128), "dense_1/bias": (128,), "dense_2/kernel": (128, 128), "dense_2/bias": (128,),
              "dense_3/kernel": (128, 32), "dense_3/bias": (32,), "global_step": (), "beta1_power": (), "beta2_power": ()}
    for k, shp in expect.items():
        assert names[k] == shp
    for v in ("dense_1/kernel", "dense_3/bias"):
        assert names[v + "/Adam"] == expect[v] and names[v + "/Adam_1"] == expect[v]
    assert len(names) == 9 + 12
    # restore into a fresh model
    m2 = xor_mlp()
    m2.compile(loss="mean_squared_error", optimizer="adam", device="cpu", seed=1)
    assert not torch.equal(m2.store.flat, m.store.flat)
    step = sv.Saver(m2.store, m2.optimizer).restore(sv.latest_checkpoint(d))
    assert step == 30
    for v, v2 in zip(m.store.trainable, m2.store.trainable):
        assert torch.equal(v.tensor, v2.tensor)


def test_property_random_bundles_roundtrip_through_both_readers(tmp_path):
    """Random checkpoints (names with shared prefixes, scalars, empty and large-ish tensors, one to three shards, int64
    and float32) survive the C++ writer -> C++ reader and C++ writer -> independent Python reader paths; enough
    entries to span several LevelDB data blocks, so restart points and index separators are exercised."""
    from hypothesis import HealthCheck, given, settings, strategies as st

    name = st.lists(st.sampled_from(["dense", "dense_1", "dense_10", "conv2d", "kernel", "bias", "Adam", "Adam_1", "a", "z"]),
                    min_size=1, max_size=4).map("/".join)
    shape = st.lists(st.integers(0, 9), min_size=0, max_size=3)
    entry = st.tuples(name, shape, st.booleans(), st.integers(0, 2))
    counter = [0]

    @settings(max_examples=25, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture])
    @given(st.lists(entry, min_size=1, max_size=120, unique_by=lambda e: e[0]), st.integers(1, 3))
    def run(entries, num_shards):
        counter[0] += 1
        prefix = str(tmp_path / ("b%d" % counter[0]) / "ckpt")
        os.makedirs(os.path.dirname(prefix), exist_ok=True)
        g = torch.Generator().manual_seed(counter[0])
        tensors = {}
        w = sv.BundleWriter(prefix, num_shards)
        for nm, shp, is_int, shard in entries:
            if is_int:
                t = torch.randint(-2 ** 40, 2 ** 40, tuple(shp), generator=g, dtype=torch.int64)
            else:
                t = torch.randn(tuple(shp), generator=g)
            tensors[nm] = t
            w.add(nm, t, shard % num_shards)
        w.finish()
        header, index = py_read_index(prefix + ".index")
        assert header.get(1, 1) == num_shards or num_shards == 1
        assert set(index) == set(tensors)
        with sv.BundleReader(prefix) as r:
            assert set(r.get_variable_to_shape_map()) == set(tensors)
            for nm, t in tensors.items():
                e = index[nm]
                assert e["shape"] == list(t.shape) and e["dtype"] == (9 if t.dtype == torch.int64 else 1)
                assert e["size"] == t.numel() * t.element_size()
                data = open("%s.data-%05d-of-%05d" % (prefix, e["shard_id"], num_shards), "rb").read()
                raw = data[e["offset"]:e["offset"] + e["size"]]
                assert mask(crc32c(raw)) == e["crc32c"]
                got = r.get_tensor(nm)
                assert got.dtype == t.dtype and torch.equal(got, t)

    run()
