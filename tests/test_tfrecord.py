"""CPU: the TensorFlow-free TFRecord reader against known answers and an independent protobuf implementation
(google.protobuf with the published tf.train.Example / TensorProto field numbers declared on the fly)."""

import struct

import numpy as np
import pytest

from dnncancerannotator_amd import tfrecord as T


def test_crc32c_known_answers():
    assert T.crc32c(b'123456789') == 0xE3069283          # the standard CRC-32C check value
    assert T.crc32c(b'') == 0
    assert T.crc32c(bytes(32)) == 0x8A9136AA             # RFC 3720 B.4: 32 bytes of zeros
    assert T.crc32c(bytes([0xFF] * 32)) == 0x62A8AB43    # RFC 3720 B.4: 32 bytes of ones
    # TFRecord mask: rotate right by 15, add 0xa282ead8
    c = T.crc32c(b'abc')
    assert T.masked_crc(b'abc') == ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


def test_native_crc32c_equals_the_table_and_payload_corruption_is_caught(tmp_path):
    """dnnca_crc32c (csrc/host_util.cpp, SSE4.2) against the byte-wise table on RFC 3720 vectors, odd lengths and unaligned starts;
    with it the reader checks every payload like tf.data.TFRecordDataset does"""
    import ctypes
    from dnncancerannotator_amd import _lib
    fn = _lib.load().dnnca_crc32c

    def native(b):
        out = ctypes.c_uint32()
        buf = np.frombuffer(b, np.uint8)
        assert fn(buf.ctypes.data if buf.size else None, buf.size, ctypes.byref(out)) == 0
        return out.value

    def table(b):
        t, c = T._crc_table(), 0xFFFFFFFF
        for x in bytes(b):
            c = int(t[(c ^ x) & 0xFF]) ^ (c >> 8)
        return c ^ 0xFFFFFFFF

    assert native(b'123456789') == 0xE3069283 and native(b'') == 0
    assert native(bytes(32)) == 0x8A9136AA and native(bytes([0xFF] * 32)) == 0x62A8AB43
    assert native(bytes(range(32))) == 0x46DD794E                        # RFC 3720 B.4: incrementing bytes
    rng = np.random.default_rng(0)
    blob = rng.integers(0, 256, 5000, dtype=np.uint8).tobytes()
    for lo, n in ((0, 5000), (1, 4099), (3, 1), (5, 7), (7, 64), (2, 1023)):
        assert native(blob[lo:lo + n]) == table(blob[lo:lo + n]), (lo, n)
        assert native(memoryview(blob)[lo:lo + n]) == table(blob[lo:lo + n])
    assert T.crc32c(blob) == table(blob)                                 # the reader's entry point takes the native path
    # a flipped payload byte is caught by default now
    s = rng.integers(0, 256, size=(2, 24, 24, 2), dtype=np.uint8)
    path = str(tmp_path / 'e.tfrecords')
    T.write_records(path, [T.make_example(s, 1, 2, '/p', 'c', ['TRA', 'label'])])
    assert len(list(T.read_records(path))) == 1
    raw = bytearray(open(path, 'rb').read())
    raw[len(raw) // 2] ^= 0x40
    bad = str(tmp_path / 'bad.tfrecords')
    open(bad, 'wb').write(bytes(raw))
    with pytest.raises(IOError, match='payload CRC'):
        list(T.read_records(bad))
    assert len(list(T.read_records(bad, verify_payload_crc=False))) == 1  # explicit opt-out
    # the reader threads check their files concurrently
    paths = []
    for i in range(6):
        p = str(tmp_path / ('t%d.tfrecords' % i))
        T.write_records(p, [T.make_example(rng.integers(0, 256, size=(3, 64, 64, 2), dtype=np.uint8), i, i, '/p', 'c', ['TRA', 'label'])])
        paths.append(p)
    for _ in range(5):
        assert [len(e) for e in T.read_exams_parallel(paths, ['TRA', 'label'], workers=4)] == [1] * 6


def _pb_messages():
    """tf.train.Example & friends + the TensorProto subset, declared with google.protobuf (independent of tfrecord.py)."""
    from google.protobuf import descriptor_pb2, descriptor_pool, message_factory
    F = descriptor_pb2.FieldDescriptorProto
    fd = descriptor_pb2.FileDescriptorProto(name='dnnca_test_example.proto', package='dt', syntax='proto3')

    def msg(name, fields):
        m = fd.message_type.add(name=name)
        for fname, num, ftype, label, tname in fields:
            f = m.field.add(name=fname, number=num, type=ftype, label=label)
            if tname:
                f.type_name = '.dt.' + tname
        return m

    OPT, REP = F.LABEL_OPTIONAL, F.LABEL_REPEATED
    msg('BytesList', [('value', 1, F.TYPE_BYTES, REP, None)])
    msg('FloatList', [('value', 1, F.TYPE_FLOAT, REP, None)])
    msg('Int64List', [('value', 1, F.TYPE_INT64, REP, None)])
    msg('Feature', [('bytes_list', 1, F.TYPE_MESSAGE, OPT, 'BytesList'), ('float_list', 2, F.TYPE_MESSAGE, OPT, 'FloatList'),
                    ('int64_list', 3, F.TYPE_MESSAGE, OPT, 'Int64List')])
    msg('FeatureEntry', [('key', 1, F.TYPE_STRING, OPT, None), ('value', 2, F.TYPE_MESSAGE, OPT, 'Feature')])
    msg('Features', [('feature', 1, F.TYPE_MESSAGE, REP, 'FeatureEntry')])       # a map is a repeated entry on the wire
    msg('Example', [('features', 1, F.TYPE_MESSAGE, OPT, 'Features')])
    msg('Dim', [('size', 1, F.TYPE_INT64, OPT, None)])
    msg('TensorShapeProto', [('dim', 2, F.TYPE_MESSAGE, REP, 'Dim')])
    msg('TensorProto', [('dtype', 1, F.TYPE_INT32, OPT, None), ('tensor_shape', 2, F.TYPE_MESSAGE, OPT, 'TensorShapeProto'),
                        ('tensor_content', 4, F.TYPE_BYTES, OPT, None)])
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    get = lambda n: message_factory.GetMessageClass(pool.FindMessageTypeByName('dt.' + n))   # noqa: E731
    return get


def test_example_wire_format_against_google_protobuf():
    get = pytest.importorskip('google.protobuf') and _pb_messages()
    rng = np.random.default_rng(0)
    slices = rng.integers(0, 256, (3, 8, 10, 4), dtype=np.uint8)
    types = ['TRA', 'ADC', 'DWI', 'label']
    # (1) what google.protobuf serialises, tfrecord.py parses
    tp = get('TensorProto')(dtype=4, tensor_content=slices.tobytes())
    for d in slices.shape:
        tp.tensor_shape.dim.add(size=d)
    ex = get('Example')()

    def add(key):
        e = ex.features.feature.add(key=key)
        return e.value
    add('slices').bytes_list.value.append(tp.SerializeToString())
    add('patientID').int64_list.value.append(1234567890123)
    add('examID').int64_list.value.append(-7)                  # negative int64: 10-byte varint
    add('path').bytes_list.value.append(b'/data/cancer/12/3')
    add('category').bytes_list.value.append(b'cancer')
    add('shape').int64_list.value.extend(slices.shape)
    add('slice_types').bytes_list.value.extend(t.encode() for t in types)
    parsed = T.parse_example(ex.SerializeToString())
    assert parsed['patientID'] == [1234567890123] and parsed['examID'] == [-7]
    assert parsed['shape'] == list(slices.shape) and parsed['slice_types'] == [t.encode() for t in types]
    assert np.array_equal(T.parse_tensor_uint8(parsed['slices'][0]), slices)
    # (2) what tfrecord.py serialises, google.protobuf parses
    mine = T.make_example(slices, 42, 7, '/p', 'healthy', types)
    back = get('Example').FromString(mine)
    feats = {e.key: e.value for e in back.features.feature}
    assert list(feats['shape'].int64_list.value) == list(slices.shape)
    assert list(feats['slice_types'].bytes_list.value) == [t.encode() for t in types]
    t2 = get('TensorProto').FromString(feats['slices'].bytes_list.value[0])
    assert t2.dtype == 4 and [d.size for d in t2.tensor_shape.dim] == list(slices.shape) and t2.tensor_content == slices.tobytes()


def test_tfrecord_dataset_element_contract(tmp_path):
    rng = np.random.default_rng(1)
    types = ['TRA', 'ADC', 'DWI', 'DCEE', 'DCEL', 'label']            # data_options.yaml:7
    exams = [rng.integers(0, 256, (n, 20, 24, 6), dtype=np.uint8) for n in (3, 2)]
    for e in exams:
        e[..., 5] = (e[..., 5] > 200) * 255                            # binary label channel
    path = str(tmp_path / 'exams.tfrecords')
    T.write_records(path, [T.make_example(e, 10 + i, i, '/x/%d' % i, 'cancer', types) for i, e in enumerate(exams)])
    # framing: length, masked crc, payload, masked crc
    raw = open(path, 'rb').read()
    n0, = struct.unpack('<Q', raw[:8])
    assert struct.unpack('<I', raw[8:12])[0] == T.masked_crc(raw[:8])
    assert struct.unpack('<I', raw[12 + n0:16 + n0])[0] == T.masked_crc(raw[12:12 + n0])
    assert len(list(T.read_records(path, verify_payload_crc=True))) == 2
    # channel selection in the requested order (slice_type_tra_dwi_adc.yaml:1), centre crop, /255, label split
    want = ['TRA', 'DWI', 'ADC', 'label']
    ds = T.TFRecordDataset([path], want, batch_size=2, output_size=(16, 16))
    batches = list(ds)
    assert [len(b[0]) for b in batches] == [2, 2, 1] and ds.element_spec[0].shape == (2, 16, 16, 3)
    x, y = batches[0]
    assert x.dtype == np.float32 and y.dtype == np.float32 and x.shape == (2, 16, 16, 3) and y.shape == (2, 16, 16)
    crop = exams[0][:2, 2:18, 4:20, :]
    assert np.array_equal(x, crop[..., [0, 2, 1]].astype(np.float32) / np.float32(255.0))
    assert np.array_equal(y, crop[..., 5].astype(np.float32) / np.float32(255.0)) and set(np.unique(y)) <= {0.0, 1.0}
    # device_convert: the same batches as uint8 RawBatches without draws; their host conversion is the float path bit for bit
    from dnncancerannotator_amd import augment
    rawds = list(T.TFRecordDataset([path], want, batch_size=2, output_size=(16, 16), device_convert=True))
    assert [len(b.raw) for b in rawds] == [2, 2, 1] and all(b.params is None and b.raw.dtype == np.uint8 for b in rawds)
    assert rawds[0].raw.shape == (2, 16, 16, 4) and rawds[0].label_index == 3
    for rb, (fx, fy) in zip(rawds, batches):
        cx, cy = augment.raw_to_float(rb)
        assert np.array_equal(cx, fx) and np.array_equal(cy, fy) and cx.flags['C_CONTIGUOUS']
    # corruption is detected
    bad = bytearray(raw)
    bad[3] ^= 0x01
    (tmp_path / 'bad.tfrecords').write_bytes(bytes(bad))
    with pytest.raises(IOError):
        list(T.read_records(str(tmp_path / 'bad.tfrecords')))
    meta = list(T.read_exams(path))
    assert meta[1].patientID == 11 and meta[1].category == 'cancer' and meta[1].slice_types == types


def test_normalize_exams_interleaves_files_equally(tmp_path):
    """data.py:517-525 / data_options.yaml:5: with normalize_exams every exam FILE contributes one slice in turn, restarting
    when it runs out, so a 6-slice exam does not outweigh a 2-slice one."""
    from dnncancerannotator_amd import tfrecord as T
    types = ['TRA', 'label']
    paths = []
    for i, n in enumerate((6, 2, 3)):
        s = np.zeros((n, 40, 40, 2), np.uint8)
        s[..., 0] = 10 * (i + 1) + np.arange(n)[:, None, None]          # slice k of file i carries the value 10 (i + 1) + k
        p = str(tmp_path / ('exam%d.tfrecords' % i))
        T.write_records(p, [T.make_example(s, i, i, '/e/%d' % i, 'cancer', types)])
        paths.append(p)
    ds = T.TFRecordDataset(paths, types, 3, output_size=(32, 32), repeat=True, drop_remainder=True, augment_options=None,
                           normalize_exams=True)
    it = iter(ds)
    seen = [int(b.raw[k, 20, 20, 0]) for b in (next(it) for _ in range(4)) for k in range(3)]
    assert seen == [10, 20, 30, 11, 21, 31, 12, 20, 32, 13, 21, 30]       # round robin; files 1 and 2 wrap around
    plain = T.TFRecordDataset(paths, types, 3, output_size=(32, 32), repeat=True, drop_remainder=True, augment_options=None)
    it = iter(plain)
    assert [int(b.raw[k, 20, 20, 0]) for b in (next(it) for _ in range(2)) for k in range(3)] == [10, 11, 12, 13, 14, 15]
    with pytest.raises(ValueError):
        T.TFRecordDataset(paths, types, 3, output_size=(32, 32), normalize_exams=True)       # needs the endless training stream


def test_a_file_listed_twice_keeps_every_file_with_its_own_exams(tmp_path):
    """The evaluation stream reads the uncached files ahead on threads and pairs results with paths: a path that occurs twice is
    read once, and neither its second occurrence nor a cache that fills up mid-pass shifts the pairing (files would otherwise be
    served -- and cached -- under their neighbour's name).  Also: a long `path` feature still decodes (only `slices` stays a view)."""
    from dnncancerannotator_amd import tfrecord as T
    types = ['TRA', 'label']
    paths = []
    for i in range(3):
        s = np.full((2, 40, 40, 2), 10 * (i + 1), np.uint8)
        p = str(tmp_path / ('exam%d.tfrecords' % i))
        T.write_records(p, [T.make_example(s, i, i, '/e/' + 'x' * 5000 + '/%d' % i, 'cancer', types)])
        paths.append(p)
    order = [paths[0], paths[1], paths[0], paths[2]]
    want = [10, 10, 20, 20, 10, 10, 30, 30]
    for cache_bytes in (8 << 30, 2 * 40 * 40 * 2 + 1, 0):           # everything cached / room for one file / no cache
        ds = T.TFRecordDataset(order, types, 2, output_size=(32, 32), device_convert=True, workers=3, cache_bytes=cache_bytes)
        for _ in range(2):                                          # second pass: from the cache where there is one
            assert [int(el.raw[k, 5, 5, 0]) for el in ds for k in range(len(el.raw))] == want
        for q, exams in ds._cache.items():
            assert int(exams[0].slices[0, 5, 5, 0]) == 10 * (paths.index(q) + 1)
    assert next(iter(T.read_exams(paths[1]))).path.endswith('/1')


def test_dataset_level_sharding_for_data_parallel(tmp_path):
    """shard=(rank, world): every rank walks the same slice stream and makes the same draws but assembles only its contiguous part
    of each global batch (engine._shard's split, before stacking / solving / uploading); evaluation remainders go to the first ranks"""
    from dnncancerannotator_amd import tfrecord as T
    rng = np.random.default_rng(3)
    paths = []
    for i, n in enumerate((5, 4)):
        s = rng.integers(0, 256, size=(n, 40, 40, 2), dtype=np.uint8)
        p = str(tmp_path / ('exam%d.tfrecords' % i))
        T.write_records(p, [T.make_example(s, i, i, '/e/%d' % i, 'cancer', ['TRA', 'label'])])
        paths.append(p)
    kw = dict(output_size=(32, 32), repeat=True, drop_remainder=True, augment_options={'random_crop': {}, 'random_flip': {}},
              buffer_size=4, seed=7)
    whole = T.TFRecordDataset(paths, ['TRA', 'label'], 4, **kw)
    parts = [T.TFRecordDataset(paths, ['TRA', 'label'], 4, shard=(r, 2), **kw) for r in range(2)]
    assert not whole.pre_sharded and all(p.pre_sharded and p.element_spec[0].shape[0] == 4 for p in parts)
    its = [iter(whole)] + [iter(p) for p in parts]
    for _ in range(5):
        w, a, b = (next(it) for it in its)
        assert np.array_equal(np.concatenate([a.raw, b.raw]), w.raw) and list(a.params) + list(b.params) == list(w.params)
        assert len(a.raw) == len(b.raw) == 2
    with pytest.raises(ValueError):
        T.TFRecordDataset(paths, ['TRA', 'label'], 3, shard=(0, 2), **kw)          # a training batch must divide evenly
    # evaluation: 9 slices in batches of 4 -> 4, 4, 1; the last batch's single slice goes to rank 0, rank 1 gets an empty batch
    for dc in (False, True):
        ev = [list(T.TFRecordDataset(paths, ['TRA', 'label'], 4, output_size=(32, 32), shard=(r, 2), device_convert=dc)) for r in range(2)]
        full = list(T.TFRecordDataset(paths, ['TRA', 'label'], 4, output_size=(32, 32), device_convert=dc))
        first = (lambda el: el.raw) if dc else (lambda el: el[0])
        assert [len(first(el)) for el in ev[0]] == [2, 2, 1] and [len(first(el)) for el in ev[1]] == [2, 2, 0]
        for f, a, b in zip(full, ev[0], ev[1]):
            assert np.array_equal(np.concatenate([first(a), first(b)]), first(f))


def test_training_batches_carry_only_the_window_the_crop_can_reach(tmp_path):
    """train_ds (data.py:62-111): the random crop moves the 16 x 16 output window by at most +-6 around the centre, so only the
    centre 28 x 28 of every 40 x 40 slice is stacked and uploaded -- and the augmentation of that window with the batch's own draws
    equals the augmentation of the whole slice (oracle/augment_oracle.py, the checker of dnnca_augment_u8)"""
    from dnncancerannotator_amd import tfrecord as T
    from oracle import augment_oracle as A
    rng = np.random.default_rng(9)
    s = rng.integers(0, 256, size=(6, 40, 40, 2), dtype=np.uint8)
    s[:, 0, 0, 0] = np.arange(6)                      # (outside every window: identifies nothing)
    s[:, 20, 20, 0] = 100 + np.arange(6)              # the slice's number sits in its centre pixel
    path = str(tmp_path / 'e.tfrecords')
    T.write_records(path, [T.make_example(s, 1, 1, '/p', 'c', ['TRA', 'label'])])
    ds = T.TFRecordDataset([path], ['TRA', 'label'], 3, output_size=(16, 16), repeat=True, drop_remainder=True,
                           augment_options={'random_crop': {}, 'random_flip': {}, 'random_contrast': {}}, seed=4)
    it = iter(ds)
    seen_jitter = set()
    for _ in range(6):
        b = next(it)
        assert b.raw.shape == (3, 28, 28, 2)
        for k in range(3):
            which = int(b.raw[k, 14, 14, 0]) - 100
            assert np.array_equal(b.raw[k], s[which, 6:34, 6:34, :])
            dy, dx, flip, contrast = b.params[k]
            seen_jitter.add((dy, dx))
            xw, yw = A.augment_image(b.raw[k], dy, dx, flip, contrast, (16, 16), 1)
            xf, yf = A.augment_image(s[which], dy, dx, flip, contrast, (16, 16), 1)
            # (the contrast mean is taken over the cropped output, so the window changes nothing)
            assert np.array_equal(xw, xf) and np.array_equal(yw, yf)
    assert len(seen_jitter) > 3 and max(max(abs(a), abs(b_)) for a, b_ in seen_jitter) <= 6
