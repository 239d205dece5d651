"""BatchFeeder: the input side of the train loop, MI355X way.

The reference hands `model.fit` a tf.data pipeline that ends in `.prefetch(AUTOTUNE)` (annotator/data.py:110,143): host threads
prepare the next batches while the device works, and Keras feeds them asynchronously (engine.py:126-135).  Here a background
thread pulls elements from the dataset iterator, takes this rank's shard and copies it into a free slot of the model's
StagingRing -- HBM staging slots filled on a copy stream (device.py, include/dnnca.h dnnca_stage_*) -- so that the upload of
batch k+1 overlaps step k on the GPU and the loop thread only enqueues steps.  The loop reads a step's scalars one step late
(StagingRing.out), which keeps one step queued behind the running one at all times.

Element kinds handed to the loop:
    ('staged', slot, x_ptr, y_ptr, n)         float32 (x, y) already on their way into `slot`
    ('raw', slot, src_ptr, RawBatch, n)        uint8 source batch of the device-side augmentation on its way into `slot`
    ('host', batch)                            anything the ring cannot take (larger than a slot): the loop uploads it itself
"""

import queue
import threading

import numpy as np

from . import augment


class BatchFeeder:
    TRAIN_SLOTS, EVAL_SLOTS = (0, 1, 2, 3), (4, 5, 6, 7)      # two feeders may work on one ring at a time: validation inside train()

    def __init__(self, device_model, iterator, shard, slots=TRAIN_SLOTS, first=None):
        """iterator: the dataset's iterator; shard(x, y=None) -> this rank's part; first: an element already drawn from it (the
        caller looked at it to size things) -- it is fed before the iterator's own elements; slots: the ring slots this feeder owns."""
        self.dm, self.it, self.shard = device_model, iterator, shard
        slot_bytes = 0
        if first is not None and isinstance(first, augment.RawBatch):
            slot_bytes = int(np.asarray(shard(first.raw)[0]).nbytes) * 5 // 4      # some room: exams differ in size
        self.ring = device_model.staging(slot_bytes=slot_bytes)
        slots = [s for s in slots if s < self.ring.slots]
        # the loop holds one slot (the step whose scalars it has not read) while it asks for the next element: with fewer than two
        # slots of its own the feeder only prefetches the elements and the loop uploads them itself
        self._stage_ok = len(slots) >= 2
        self.free = queue.Queue()
        for s in slots:
            self.free.put(s)
        self.ready = queue.Queue(maxsize=max(1, len(slots)))      # bounded even without slots of its own: 'host' elements are whole batches
        self._first = first
        self._stop = False
        self.thread = threading.Thread(target=self._run, name='dnnca-batch-feeder', daemon=True)
        self.thread.start()

    # ---- producer thread ------------------------------------------------------------------------------------------
    def _elements(self):
        if self._first is not None:
            yield self._first
            self._first = None
        for el in self.it:
            yield el

    def _stage(self, batch):
        if not self._stage_ok:
            return ('host', batch)
        if isinstance(batch, augment.RawBatch):
            raw = np.ascontiguousarray(self.shard(batch.raw)[0], np.uint8)
            # a batch the model cannot take as it is -- too large for a slot or for max_batch, or cropped to another size than the
            # built input -- goes to the loop unstaged: its host path chunks it or raises the shape error (never a silent misread)
            if (not len(raw) or not self.ring.fits(raw) or len(raw) > self.dm.max_batch or raw.ndim != 4 or
                    tuple(int(v) for v in batch.output_size) != tuple(self.dm.in_shape[:2]) or raw.shape[-1] - 1 != self.dm.in_shape[2]):
                return ('host', batch)
            slot = self.free.get()
            if slot is None:
                return None
            src, _ = self.ring.upload(slot, raw)
            return ('raw', slot, src, batch, len(raw))
        x, y = self.shard(np.asarray(batch[0]), np.asarray(batch[1]))
        x, y = np.ascontiguousarray(x, np.float32), np.ascontiguousarray(y, np.float32)
        if (not len(x) or not self.ring.fits(x, y) or len(x) > self.dm.max_batch or x.ndim != 4 or
                tuple(x.shape[1:]) != tuple(self.dm.in_shape) or y.shape != x.shape[:3]):
            return ('host', batch)        # (a shape the built model does not have: DeviceModel._check_x raises on the host path)
        slot = self.free.get()
        if slot is None:
            return None
        px, py = self.ring.upload(slot, x, y)
        return ('staged', slot, px, py, len(x))

    def _run(self):
        try:
            for batch in self._elements():
                if self._stop:
                    break
                item = self._stage(batch)
                if item is None or self._stop:
                    break
                self.ready.put(item)
        except BaseException as e:          # surfaces in the loop thread, at the position of the failing element
            self.ready.put(('error', e))
        self.ready.put(('end',))

    # ---- loop thread ------------------------------------------------------------------------------------------------
    def __iter__(self):
        return self

    def __next__(self):
        item = self.ready.get()
        if item[0] == 'end':
            self.ready.put(item)            # stays exhausted
            raise StopIteration
        if item[0] == 'error':
            raise item[1]
        return item

    def release(self, slot):
        """the slot's step has been enqueued and its outputs read: the producer may fill it again"""
        self.free.put(slot)

    def close(self):
        self._stop = True
        self.free.put(None)
        while self.thread.is_alive():       # unblock a producer waiting on a full `ready` queue
            try:
                self.ready.get_nowait()
            except queue.Empty:
                pass
            self.thread.join(0.01)
