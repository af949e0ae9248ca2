"""Host glue used by the training driver (style/utils/misc.py:10-82,114-134): file listing, the
progress meter with the reference's momentum averaging, small dict helpers."""
import glob
import math
import os


def iter_all_files(path, pattern='**/*'):
    for name in glob.iglob(os.path.join(path, pattern), recursive=True):
        if os.path.isfile(name):
            yield name


def dict_map(func, d, recursive=False):
    if recursive and isinstance(d, dict):
        return {k: dict_map(func, v, True) for k, v in d.items()}
    if isinstance(d, dict):
        return {k: func(v) for k, v in d.items()}
    return func(d)


def flatten_underscore(d, prefix=''):
    """flatten_dict(d, reducer='underscore') for the nested loss dict (train-model.py:148)."""
    out = {}
    for k, v in d.items():
        if isinstance(v, dict):
            out.update(flatten_underscore(v, prefix + k + '_'))
        else:
            out[prefix + k] = v
    return out


def _hashable(obj):
    if isinstance(obj, (list, tuple)):
        return tuple(_hashable(o) for o in obj)
    return frozenset(obj) if isinstance(obj, set) else obj


def group_by(data, key=None, attr=None, func=None, save_indices=False):
    """Buckets of `data` in first-seen key order (style/utils/misc.py:93-113, used by style/data.py:69): the key is a
    callable, an item name (`key='name'`), an attribute name (`attr=`) or the element itself; `save_indices` collects
    positions instead of elements; `func` maps every bucket."""
    if callable(key):
        key_of = key
    elif key:
        key_of = lambda x: x[key]                # noqa: E731
    elif attr:
        key_of = lambda x: getattr(x, attr)      # noqa: E731
    else:
        key_of = lambda x: x                     # noqa: E731
    buckets = {}
    for i, elem in enumerate(data):
        buckets.setdefault(_hashable(key_of(elem)), []).append(i if save_indices else elem)
    return {k: func(v) for k, v in buckets.items()} if func else buckets


def flatten(elems):
    """One level of nesting removed (style/utils/misc.py:116-117)."""
    return [x for sub in elems for x in sub]


def make_dirs(path):
    os.makedirs(path or '.', exist_ok=True)


def assert_dir(path):
    make_dirs(os.path.dirname(path))


class ProgressBar:
    """Momentum-averaged values shown behind a tqdm bar (style/utils/misc.py:17-82).  Unbiased form:
    avg_k = S_k / N_k with S_k <- S_k*m + v*n and N_k <- N_k*m + n; biased (after `initial_values`):
    avg_k <- avg_k*m + v*(1-m).  Falls back to plain printing when tqdm is not installed."""

    def __init__(self, n_iterations=None, momentum=.99, biased=False, show_min_for=(), show_max_for=()):
        self.n_iterations, self.momentum, self.biased = n_iterations, momentum, biased
        self.show_min_for, self.show_max_for = show_min_for, show_max_for
        self.decimal_places = 2
        self.n = 0
        try:
            from tqdm import tqdm
            self.pbar = tqdm(total=n_iterations)
        except ImportError:
            self.pbar = None
        self.clear_values()

    def clear_values(self):
        self.values_sum, self.values_seen = {}, {}
        self.min_values, self.max_values, self.avg_values = {}, {}, {}

    def initial_values(self, **values):
        self.avg_values.update(values)
        self.biased = True

    def add(self, n, **values):
        self.n += n
        if self.pbar is not None:
            self.pbar.update(n)
        self.update_values(n, **values)
        if self.n == self.n_iterations:
            self.close()

    def update_values(self, n, **values):
        values = {k: v for k, v in values.items() if v is not None}
        m = self.momentum
        if self.biased:
            self.avg_values.update({k: self.avg_values.get(k, 0) * m + v * (1 - m) for k, v in values.items()})
        else:
            for k, v in values.items():
                self.values_sum[k] = self.values_sum.get(k, 0) * m + v * n
                self.values_seen[k] = self.values_seen.get(k, 0) * m + n
            self.avg_values = {k: self.values_sum[k] / self.values_seen[k] for k in self.values_sum}
        self.min_values = {k: min(avg, self.min_values.get(k) or math.inf) for k, avg in self.avg_values.items()}
        self.max_values = {k: max(avg, self.max_values.get(k) or -math.inf) for k, avg in self.avg_values.items()}
        if self.pbar is not None:
            self.pbar.set_postfix_str(self.description())

    def description(self):
        p = self.decimal_places
        parts = [f'{k}: {v:.{p}f}' for k, v in self.avg_values.items()]
        parts += [f'min {k}: {v:.{p}f}' for k, v in self.min_values.items() if k in self.show_min_for]
        parts += [f'max {k}: {v:.{p}f}' for k, v in self.max_values.items() if k in self.show_max_for]
        return ', '.join(parts)

    def close(self):
        if self.pbar is not None:
            self.pbar.close()

    def __getitem__(self, k):
        return self.avg_values[k]
