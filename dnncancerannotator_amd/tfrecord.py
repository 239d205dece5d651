"""Reader (and writer, for tests / synthetic data) of the reference's TFRecord exam files -- without TensorFlow.

Schema (annotator/data.py:238-254 writer, :448-470 reader): uncompressed TFRecord framing (`_TFRECORD_COMPRESSION = None`,
data.py:59); each record is a `tf.train.Example` with features
    slices      bytes   tf.io.serialize_tensor(uint8 [N, H, W, C])  = a serialized TensorProto
    patientID   int64, examID int64, path bytes, category bytes, shape int64[4], slice_types bytes list (C names).
The element pipeline on top of it mirrors data.py:473-487 (channel selection by slice type), :195-206 (centre crop,
/255) and :766-788 (`label` channel -> y, the others -> x).

Formats implemented from their public specifications: TFRecord framing (length, masked CRC-32C, data, masked CRC-32C),
protobuf wire format (varint / 64-bit / length-delimited / 32-bit), tensorflow/core/example/{example,feature}.proto and
tensorflow/core/framework/{tensor,tensor_shape}.proto field numbers.  No file of the reference ships with the repo."""

import os
import struct
from collections import namedtuple

import numpy as np

# ------------------------------------------------------------------------------------------------ CRC-32C (Castagnoli)
_CRC_TABLE = None


def _crc_table():
    global _CRC_TABLE
    if _CRC_TABLE is None:
        poly = 0x82F63B78
        t = np.zeros(256, np.uint32)
        for i in range(256):
            c = i
            for _ in range(8):
                c = (c >> 1) ^ poly if c & 1 else c >> 1
            t[i] = c
        _CRC_TABLE = t
    return _CRC_TABLE


_native_crc = None


def _native():
    """dnnca_crc32c of libdnnca (csrc/host_util.cpp: the SSE4.2 crc32 instruction), or False when the library is not built"""
    global _native_crc
    if _native_crc is None:
        try:
            import ctypes
            from . import _lib
            fn = _lib.load().dnnca_crc32c

            def crc(data):
                out = ctypes.c_uint32()                           # per call: the reader threads check their files concurrently
                buf = np.frombuffer(data, np.uint8)              # bytes, bytearray or a (read-only) memoryview: no copy
                if fn(buf.ctypes.data, buf.size, ctypes.byref(out)) != 0:
                    raise ValueError('dnnca_crc32c failed')
                return out.value
            _native_crc = crc
        except Exception:
            _native_crc = False
    return _native_crc


def crc32c(data):
    native = _native()
    if native and len(data) > 64:
        return native(data)
    t = _crc_table()
    c = 0xFFFFFFFF
    for b in bytes(data):
        c = int(t[(c ^ b) & 0xFF]) ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def masked_crc(data):
    c = crc32c(data)
    return ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


# ------------------------------------------------------------------------------------------------ TFRecord framing
def read_records(path, verify_payload_crc=None):
    """Yields the payload of every record as a memoryview into a read-only mapping of the file (no copy: an exam's pixels go from
    the page cache straight into whoever picks them apart).  The 12-byte header CRC is always checked; the payload CRC -- as
    tf.data.TFRecordDataset does -- whenever libdnnca's dnnca_crc32c is there to walk the megabytes of pixels (None), or on
    request (True: with the byte-wise Python loop if need be)."""
    import mmap
    if verify_payload_crc is None:
        verify_payload_crc = bool(_native())
    size = os.path.getsize(path)
    if size == 0:
        return
    with open(path, 'rb') as f:
        mm = mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ)      # stays alive as long as a payload view does
    buf = memoryview(mm)
    pos = 0
    while pos < size:
        if size - pos < 12:
            raise IOError('truncated TFRecord header in %s' % path)
        header = bytes(buf[pos:pos + 12])
        length, = struct.unpack('<Q', header[:8])
        if struct.unpack('<I', header[8:])[0] != masked_crc(header[:8]):
            raise IOError('corrupt TFRecord length CRC in %s' % path)
        if size - pos - 12 < length + 4:
            raise IOError('truncated TFRecord payload in %s' % path)
        data = buf[pos + 12:pos + 12 + length]
        if verify_payload_crc and struct.unpack('<I', bytes(buf[pos + 12 + length:pos + 16 + length]))[0] != masked_crc(data):
            raise IOError('corrupt TFRecord payload CRC in %s' % path)
        pos += 16 + length
        yield data


def write_records(path, payloads):
    with open(path, 'wb') as f:
        for data in payloads:
            head = struct.pack('<Q', len(data))
            f.write(head + struct.pack('<I', masked_crc(head)) + data + struct.pack('<I', masked_crc(data)))


# ------------------------------------------------------------------------------------------------ protobuf wire format
def _varint(buf, pos):
    result, shift = 0, 0
    while True:
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not b & 0x80:
            return result, pos
        shift += 7


def _fields(buf):
    """Yields (field number, wire type, value) of one message; length-delimited values are memoryviews."""
    buf = memoryview(buf)
    pos, n = 0, len(buf)
    while pos < n:
        key, pos = _varint(buf, pos)
        num, wt = key >> 3, key & 7
        if wt == 0:
            val, pos = _varint(buf, pos)
        elif wt == 1:
            val, pos = bytes(buf[pos:pos + 8]), pos + 8
        elif wt == 2:
            ln, pos = _varint(buf, pos)
            val, pos = buf[pos:pos + ln], pos + ln
        elif wt == 5:
            val, pos = bytes(buf[pos:pos + 4]), pos + 4
        else:
            raise ValueError('unsupported protobuf wire type %d' % wt)
        yield num, wt, val


def _enc_varint(v):
    out = bytearray()
    v &= (1 << 64) - 1
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _enc_field(num, wt, payload):
    if wt == 2:
        return _enc_varint((num << 3) | 2) + _enc_varint(len(payload)) + bytes(payload)
    return _enc_varint((num << 3) | wt) + bytes(payload)


# ------------------------------------------------------------------------------------------------ Example / TensorProto
def parse_example(data):
    """tf.train.Example -> {name: list of bytes | list of int | list of float}."""
    out = {}
    for num, _, features in _fields(data):
        if num != 1:
            continue
        for fnum, _, entry in _fields(features):           # map<string, Feature> entries
            if fnum != 1:
                continue
            key, feature = None, None
            for enum, _, v in _fields(entry):
                if enum == 1:
                    key = bytes(v).decode()
                elif enum == 2:
                    feature = v
            values = []
            for knum, _, lst in _fields(feature if feature is not None else b''):
                for vnum, vwt, v in _fields(lst):
                    if vnum != 1:
                        continue
                    if knum == 1:                           # BytesList (a serialized tensor stays a view: megabytes of pixels)
                        values.append(v if key == 'slices' and len(v) > 4096 else bytes(v))      # only the pixels stay a view
                    elif knum == 3:                         # Int64List (packed or not)
                        if vwt == 2:
                            pos, raw = 0, bytes(v)
                            while pos < len(raw):
                                x, pos = _varint(raw, pos)
                                values.append(x - (1 << 64) if x >= 1 << 63 else x)
                        else:
                            values.append(v - (1 << 64) if v >= 1 << 63 else v)
                    elif knum == 2:                         # FloatList
                        raw = bytes(v)
                        values.extend(struct.unpack('<%df' % (len(raw) // 4), raw))
            out[key] = values
    return out


DT_UINT8 = 4


def parse_tensor_uint8(data):
    """tf.io.parse_tensor(x, tf.uint8) for what tf.io.serialize_tensor wrote: dtype 1, tensor_shape 2, tensor_content 4."""
    dtype, dims, content = None, [], None
    for num, _, v in _fields(data):
        if num == 1:
            dtype = v
        elif num == 2:
            for snum, _, dim in _fields(v):
                if snum == 2:
                    for dnum, _, size in _fields(dim):
                        if dnum == 1:
                            dims.append(size)
        elif num == 4:
            content = v
    if dtype != DT_UINT8:
        raise ValueError('expected a uint8 tensor (dtype %d), got dtype %s' % (DT_UINT8, dtype))
    return np.frombuffer(content, np.uint8).reshape(dims)


def serialize_tensor_uint8(a):
    a = np.ascontiguousarray(a, np.uint8)
    shape = b''.join(_enc_field(2, 2, _enc_field(1, 0, _enc_varint(d))) for d in a.shape)
    return _enc_field(1, 0, _enc_varint(DT_UINT8)) + _enc_field(2, 2, shape) + _enc_field(4, 2, a.tobytes())


def make_example(slices, patient_id, exam_id, path, category, slice_types):
    """The record annotator/data.py:238-254 writes for one exam (slices uint8 [N, H, W, C])."""
    def bytes_list(vals):
        return _enc_field(1, 2, b''.join(_enc_field(1, 2, v) for v in vals))

    def int64_list(vals):
        return _enc_field(3, 2, _enc_field(1, 2, b''.join(_enc_varint(int(v)) for v in vals)))

    feats = {
        'slices': bytes_list([serialize_tensor_uint8(slices)]),
        'patientID': int64_list([patient_id]),
        'examID': int64_list([exam_id]),
        'path': bytes_list([path.encode()]),
        'category': bytes_list([category.encode()]),
        'shape': int64_list(slices.shape),
        'slice_types': bytes_list([s.encode() for s in slice_types]),
    }
    entries = b''.join(_enc_field(1, 2, _enc_field(1, 2, k.encode()) + _enc_field(2, 2, v)) for k, v in feats.items())
    return _enc_field(1, 2, entries)


# ------------------------------------------------------------------------------------------------ dataset
Exam = namedtuple('Exam', ['slices', 'patientID', 'examID', 'path', 'category', 'slice_types'])
Spec = namedtuple('Spec', ['shape', 'dtype'])


def read_exams(path, output_slice_types=None):
    """extract_slices_from_tfrecord (data.py:438-512) up to the per-exam level: selects / orders the channels named in
    `output_slice_types` (data.py:473-487)."""
    for data in read_records(path):
        ex = parse_example(data)
        shape = ex['shape']
        slices = parse_tensor_uint8(ex['slices'][0]).reshape(shape)
        types = [t.decode() for t in ex['slice_types']]
        if output_slice_types is not None:
            idx = [types.index(t) for t in output_slice_types]
            if idx != list(range(slices.shape[-1])):
                # channel by channel into a C-ordered array (`slices[..., idx]` would come back with the channel axis outermost in
                # memory: every later pass over it -- crop, cast, stack -- then crawls)
                picked = np.empty(slices.shape[:-1] + (len(idx),), np.uint8)
                for j, i in enumerate(idx):
                    picked[..., j] = slices[..., i]
                slices = picked
            types = list(output_slice_types)
        yield Exam(slices, ex['patientID'][0], ex['examID'][0], ex['path'][0].decode(), ex['category'][0].decode(), types)


def read_exams_parallel(paths, output_slice_types=None, workers=None):
    """[exams of paths[0]], [exams of paths[1]], ... in order, with up to `workers` files being read and decoded ahead on threads --
    what `interleave(..., num_parallel_calls=AUTOTUNE)` / the parallel maps of annotator/data.py:178,283,294 do for the
    reference.  File reads, the big byte copies and the channel picks all release the GIL."""
    paths = list(paths)
    if workers is None:
        workers = min(8, max(1, (os.cpu_count() or 2) // 2))
    if workers <= 1 or len(paths) <= 1:
        for p in paths:
            yield list(read_exams(p, output_slice_types))
        return
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(workers, thread_name_prefix='dnnca-exam-reader') as pool:
        pending, nxt = [], 0
        while pending or nxt < len(paths):
            while nxt < len(paths) and len(pending) < workers + 1:
                pending.append(pool.submit(lambda q: list(read_exams(q, output_slice_types)), paths[nxt]))
                nxt += 1
            yield pending.pop(0).result()


class TFRecordDataset:
    """Batches from the reference's .tfrecords exam files.

    Evaluation (`augment_options` False): (x float32 [B, H, W, C] in [0, 1], y float32 [B, H, W]) -- centre crop to `output_size`
    (data.py:195-200), /255 (data.py:205-206), `label` channel -> y, the remaining channels -> x in `slice_types` order
    (data.py:766-788), all on the host.

    Training (`augment_options` a dict or None, data.py:62-111 train_ds): the slices are centre-cropped to 512 x 512 (the `base`
    call of train_ds, data.py:95-100), shuffled through a buffer of `buffer_size` slices (data.py:106) and handed on as uint8
    `augment.RawBatch`es with their random draws; crop / flip / contrast / the /255 and the feature-label split then run on the
    device (`dnnca_augment_u8` / `dnnca_warp_f32`, engine.train)."""

    def __init__(self, paths, slice_types, batch_size, output_size=(512, 512), repeat=False, drop_remainder=False,
                 augment_options=False, buffer_size=0, seed=0, normalize_exams=False, device_convert=False, workers=None,
                 cache_bytes=8 << 30, shard=None, **ignored):
        from . import augment
        self.paths = list(paths)
        self.slice_types = list(slice_types)
        assert 'label' in self.slice_types, 'slice_types must name the label channel (data.py:771)'
        self.batch_size, self.output_size = int(batch_size), tuple(output_size)
        self.repeat, self.drop_remainder = repeat, drop_remainder
        self.feature_idx = [i for i, t in enumerate(self.slice_types) if t != 'label']
        self.label_idx = self.slice_types.index('label')
        self.plan = None if augment_options is False else augment.parse_augment_options(augment_options, self.output_size)
        if self.plan is not None:
            self.output_size = self.plan.output_size
        self.buffer_size = int(buffer_size)
        self.workers = workers                # reader threads (None: half the cores, at most eight)
        # data parallel, one process per GPU: shard = (rank, world) makes every batch this rank's contiguous part of the global
        # batch (engine._shard's split, done BEFORE the slices are stacked, solved for and uploaded -- every rank walks the same
        # slice stream and makes the same draws, but only pays for its own part).  element_spec keeps the global batch size.
        self.rank, self.world = (int(shard[0]), int(shard[1])) if shard else (0, 1)
        self.pre_sharded = self.world > 1
        if self.pre_sharded and augment_options is not False and self.batch_size % self.world:
            raise ValueError('global batch %d is not divisible by %d ranks' % (self.batch_size, self.world))
        # decoded exams (uint8, channels picked) stay in host memory up to `cache_bytes`: the endless training stream re-reads
        # every file each time its slices run out (data.py:517-525), and `annotator evaluate` walks the files once per checkpoint
        self.cache_bytes = int(cache_bytes)
        self._cache, self._cached_bytes = {}, 0
        self._pool, self._ahead, self._index = None, {}, {p: i for i, p in enumerate(self.paths)}
        # evaluation with device_convert: the centre-cropped uint8 slices travel as `augment.RawBatch`es without draws (params None)
        # -- a quarter of the float bytes over PCIe, no float copy of an exam on the host; the engine converts them on the device
        # (or, without one, with augment.raw_to_float)
        self.device_convert = bool(device_convert) and self.plan is None
        # data.py:517-525 (base_from_tfrecords, normalize=True; data_options.yaml:5 for training): the files are interleaved one
        # slice at a time, each file's slice stream repeated for ever, so that every exam file contributes equally however many
        # slices it holds.  (tf.data's interleave only ever opens `cycle_length` = #cores files when the streams are infinite;
        # here the round robin runs over ALL files, which is what the option is documented to mean, data.py:81.)
        self.normalize_exams = bool(normalize_exams)
        if self.normalize_exams and not (self.repeat and self.plan is not None):
            raise ValueError('normalize_exams makes an endless training stream: it needs repeat=True and augment_options')
        self.rng = np.random.default_rng(seed)
        self.element_spec = (Spec((self.batch_size,) + self.output_size + (len(self.feature_idx),), np.float32),
                             Spec((self.batch_size,) + self.output_size, np.float32))

    def _exams_of(self, path, exams=None):
        """the decoded exams of one file, from the cache when they are there (`exams`: just read by the caller -> remember them).
        A miss reads the file on a reader thread and, while at it, starts the following uncached files of `paths` as well (the
        round robin of normalize_exams asks for them in that order): the first pass over a data set runs on `workers` threads."""
        got = self._cache.get(path)
        if got is not None:
            return got
        if exams is None:
            workers = self.workers if self.workers is not None else min(8, max(1, (os.cpu_count() or 2) // 2))
            if workers <= 1:
                exams = list(read_exams(path, self.slice_types))
            else:
                if self._pool is None:
                    from concurrent.futures import ThreadPoolExecutor
                    self._pool = ThreadPoolExecutor(workers, thread_name_prefix='dnnca-exam-reader')
                read = lambda q: list(read_exams(q, self.slice_types))      # noqa: E731
                if path not in self._ahead:
                    self._ahead[path] = self._pool.submit(read, path)
                i = self._index.get(path, 0)
                for q in self.paths[i + 1:i + 1 + workers]:                 # read-ahead window behind the file asked for
                    if len(self._ahead) > workers:
                        break
                    if q not in self._cache and q not in self._ahead:
                        self._ahead[q] = self._pool.submit(read, q)
                exams = self._ahead.pop(path).result()
        size = sum(e.slices.nbytes for e in exams)
        if self._cached_bytes + size <= self.cache_bytes:
            self._cache[path] = exams
            self._cached_bytes += size
        return exams

    def _exam_lists(self):
        """[exams of file 0], [exams of file 1], ... -- cached files at once, the others through the reader threads"""
        # the uncached files are read ahead once each (a path listed twice is read once); results are looked up by PATH, so a
        # file that entered the cache meanwhile (its second occurrence, another iterator) never shifts the pairing
        missing = list(dict.fromkeys(p for p in self.paths if p not in self._cache))
        fresh = zip(missing, read_exams_parallel(missing, self.slice_types, self.workers))
        got = {}
        for p in self.paths:
            if p in self._cache:
                yield self._exams_of(p)
                continue
            while p not in got:
                q, exams = next(fresh)
                got[q] = exams
            exams = self._exams_of(p, got[p])
            if p in self._cache:
                del got[p]                       # cached now: a later occurrence comes from there
            yield exams

    def _mine(self, items):
        """this rank's contiguous part of a global batch (a remainder goes to the first ranks: no evaluation sample is dropped)"""
        if self.world == 1:
            return items
        from .distributed import shard_bounds
        lo, hi = shard_bounds(len(items), self.rank, self.world, even=False)
        return items[lo:hi]

    @staticmethod
    def _centre(s, oh, ow):
        gy, gx = (s.shape[1] - oh) // 2, (s.shape[2] - ow) // 2
        return s[:, gy:gy + oh, gx:gx + ow, :]

    def _file_slices(self, path):
        for exam in self._exams_of(path):
            s = exam.slices
            s = self._centre(s, min(512, s.shape[1]), min(512, s.shape[2]))
            for k in range(len(s)):
                yield s[k]

    def _raw_slices(self):
        """uint8 [H, W, Cs] slices, centre-cropped like train_ds's base() call (512 x 512, data.py:97): in file order, or --
        normalize_exams -- one slice from each file in turn, every file restarting when it runs out (endless)."""
        if not self.normalize_exams:
            for exams in self._exam_lists():
                for exam in exams:
                    s = self._centre(exam.slices, min(512, exam.slices.shape[1]), min(512, exam.slices.shape[2]))
                    for k in range(len(s)):
                        yield s[k]
            return
        streams = [self._file_slices(p) for p in self.paths]
        empty = set()
        while len(empty) < len(streams):
            for i, path in enumerate(self.paths):
                if i in empty:
                    continue
                try:
                    yield next(streams[i])
                except StopIteration:
                    streams[i] = self._file_slices(path)
                    try:
                        yield next(streams[i])
                    except StopIteration:
                        empty.add(i)          # a file without slices drops out of the rotation

    def _shuffled(self, it):
        """tf.data shuffle(buffer_size): fill a buffer, then emit a random element and replace it with the next one."""
        if self.buffer_size <= 1:
            yield from it
            return
        buf = []
        for el in it:
            if len(buf) < self.buffer_size:
                buf.append(el)
                continue
            k = int(self.rng.integers(len(buf)))
            out, buf[k] = buf[k], el
            yield out
        while buf:
            k = int(self.rng.integers(len(buf)))
            buf[k], buf[-1] = buf[-1], buf[k]
            yield buf.pop()

    def _slices(self):
        oh, ow = self.output_size
        for exams in self._exam_lists():
            for exam in exams:
                s = self._centre(exam.slices, oh, ow).astype(np.float32) / np.float32(255.0)
                for k in range(len(s)):
                    yield s[k][..., self.feature_idx], s[k][..., self.label_idx]

    def _augmented(self):
        from . import augment
        raws = []
        for r in self._shuffled(self._raw_slices()):
            raws.append(r)
            if len(raws) == self.batch_size:
                yield self._raw_batch(raws)
                raws = []
        if raws and not self.drop_remainder:
            yield self._raw_batch(raws)

    def _raw_batch(self, raws):
        from . import augment
        warp = None
        if self.plan.warp is not None:
            if self.output_size[0] != self.output_size[1]:
                raise ValueError('random_warp supports square images only (data.py:746 asserts width == height)')
            src, dst = augment.draw_warp(self.rng, len(raws), self.output_size[0], **self.plan.warp)      # draws for the global batch
            warp = augment.solve_warp(self._mine(src), self._mine(dst))
        params = self._mine(augment.draw_params(self.rng, len(raws), self.plan))
        # only the window the random crop can reach travels on (centre +- the jitter bound: the same centre, so the same pixels
        # come out of dnnca_augment_u8): 268 x 268 of 512 x 512 for the default crop -- a quarter of the bytes to stack and upload
        m = max(abs(int(self.plan.crop['min_'])), abs(int(self.plan.crop['max_']))) if self.plan.crop is not None else 0
        oh, ow = self.output_size
        mine = [self._centre(r[None], min(r.shape[0], oh + 2 * m), min(r.shape[1], ow + 2 * m))[0] for r in self._mine(raws)]
        return augment.RawBatch(np.stack(mine), params, self.output_size, self.label_idx, warp)

    def _stacked(self, xs, ys):
        mx, my = self._mine(xs), self._mine(ys)
        if not mx:
            return np.zeros((0,) + xs[0].shape, np.float32), np.zeros((0,) + ys[0].shape, np.float32)
        return np.stack(mx), np.stack(my)

    def _eval_raw(self):
        from . import augment
        oh, ow = self.output_size
        raws = []

        def batch():
            mine = self._mine(raws)
            raw = np.stack(mine) if mine else np.zeros((0,) + raws[0].shape, np.uint8)
            return augment.RawBatch(raw, None, self.output_size, self.label_idx, None)

        for exams in self._exam_lists():
            for exam in exams:
                s = self._centre(exam.slices, oh, ow)
                for k in range(len(s)):
                    raws.append(s[k])
                    if len(raws) == self.batch_size:
                        yield batch()
                        raws = []
        if raws and not self.drop_remainder:
            yield batch()

    def __iter__(self):
        while True:
            if self.plan is not None:
                yield from self._augmented()
            elif self.device_convert:
                yield from self._eval_raw()
            else:
                xs, ys = [], []
                for x, y in self._slices():
                    xs.append(x)
                    ys.append(y)
                    if len(xs) == self.batch_size:
                        yield self._stacked(xs, ys)
                        xs, ys = [], []
                if xs and not self.drop_remainder:
                    yield self._stacked(xs, ys)
            if not self.repeat:
                return
