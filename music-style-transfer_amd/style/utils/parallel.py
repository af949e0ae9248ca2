"""One-thread prefetch of the host data pipeline (style/utils/parallel.py:6-76): the MIDI parsing of
song i+1 overlaps the GPU iteration on song i.  Exceptions raised by the producer are re-raised in
the consumer."""
import queue
import threading

_END = object()


class ParallelIterable:
    def __init__(self, iterator, n_jobs=1, max_queue_size=1):
        self.iterator = iterator
        self.queue = queue.Queue(maxsize=max_queue_size)
        self.lock = threading.Lock()
        self.stopped = threading.Event()
        self.exhausted = False
        self.threads = [threading.Thread(target=self._produce, daemon=True) for _ in range(n_jobs)]
        for t in self.threads:
            t.start()

    def _produce(self):
        while not self.stopped.is_set():
            with self.lock:                       # the generator is shared: one producer advances it at a time
                if self.exhausted:
                    return
                try:
                    item = (True, next(self.iterator))
                except StopIteration:
                    self.exhausted, item = True, (True, _END)
                except Exception as e:            # handed to the consumer
                    self.exhausted, item = True, (False, e)
            while not self.stopped.is_set():
                try:
                    self.queue.put(item, timeout=.1)
                    break
                except queue.Full:
                    pass

    def __iter__(self):
        while True:
            ok, value = self.queue.get()
            if not ok:
                raise value
            if value is _END:
                return
            yield value

    def stop(self, timeout=None):
        self.stopped.set()
        for t in self.threads:
            t.join(timeout)


def iter_parallel(iterable, *args, **kwargs):
    return iter(ParallelIterable(iterable, *args, **kwargs))
